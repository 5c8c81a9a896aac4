'use strict';
// ORACLE / TEST INFRASTRUCTURE — command-line front end used by tests/ and by bench.py's
// cpu_baseline leg.  Never used by the product path.
//
//   node oracle/cli.js restate   <scene.json> <w> <h> [row0 row1] [--out f.rgba]   JS restatement
//   node oracle/cli.js reference <scene.json> <w> <h> [row0 row1] [--out f.rgba]   /root/reference/main.js (build container only)
//   node oracle/cli.js main      <w> <h> [--out f.rgba]                            the reference's own main()
//   node oracle/cli.js flatten   <scene.json> --out f.blob                         js/flatten.js blob (host-logic parity with Python)
//   node oracle/cli.js time      <scene.json> <w> <h> <rows> [reps]                warm-up + timed restatement of <rows> rows, reps times (1 thread)
// Prints one JSON line with sha256 / counters / timings.

const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const F = require('../html5-canvas-raytracer_amd/js/flatten.js');

const argv = process.argv.slice(2);
const outIdx = argv.indexOf('--out');
const outFile = outIdx >= 0 ? argv[outIdx + 1] : null;
const args = outIdx >= 0 ? argv.slice(0, outIdx).concat(argv.slice(outIdx + 2)) : argv;
const cmd = args[0];
const sha = (b) => crypto.createHash('sha256').update(Buffer.from(b)).digest('hex');
const loadScene = (p) => F.sceneFromJSON(fs.readFileSync(p, 'utf8'), path.dirname(p));
const emit = (o) => console.log(JSON.stringify(o));

if (cmd === 'restate' || cmd === 'reference') {
  const scene = loadScene(args[1]);
  const w = +args[2], h = +args[3];
  const row0 = args[4] === undefined ? 0 : +args[4], row1 = args[5] === undefined ? h : +args[5];
  const t0 = process.hrtime.bigint();
  let r;
  if (cmd === 'restate') r = require('./restate.js').render(w, h, scene, {row0, row1});
  else {
    const H = require('./ref_harness.js');
    if (!H.available()) { emit({error: 'reference not available'}); process.exit(3); }
    const k = scene.supersample || 1;
    if (k > 1) {
      const plain = Object.assign({}, scene, {supersample: 1});
      const hi = H.renderScene(plain, k * w, k * h, {row0: k * row0, row1: k * row1});
      r = {rgba: H.boxFilter(hi.rgba, k * w, k * (row1 - row0), k)};
    } else r = H.renderScene(scene, w, h, {row0, row1});
  }
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  if (outFile) fs.writeFileSync(outFile, Buffer.from(r.rgba));
  emit({cmd, w, h, row0, row1, sha256: sha(r.rgba), ms, rays: r.rays, shadowRays: r.shadowRays, sphereTests: r.sphereTests});
} else if (cmd === 'main') {
  const H = require('./ref_harness.js');
  if (!H.available()) { emit({error: 'reference not available'}); process.exit(3); }
  const rgba = H.runMain(+args[1], +args[2]);
  if (outFile) fs.writeFileSync(outFile, Buffer.from(rgba));
  emit({cmd, w: +args[1], h: +args[2], sha256: sha(rgba)});
} else if (cmd === 'flatten') {
  const blob = Buffer.from(F.flattenScene(loadScene(args[1])));
  if (outFile) fs.writeFileSync(outFile, blob);
  emit({cmd, bytes: blob.length, sha256: sha(blob)});
} else if (cmd === 'time') {
  // Single-thread CPU baseline: one untimed warm-up pass (JIT), then `rows` rows spread evenly
  // over the frame (so sky, spheres and floor are sampled in proportion); the caller scales by
  // rows (SURVEY §8(d) timing protocol).
  const scene = loadScene(args[1]);
  const w = +args[2], h = +args[3], rows = Math.min(+args[4], h);
  const R = require('./restate.js');
  const reps = Math.max(1, +(args[5] || 1));
  const list = []; for (let k = 0; k < rows; k++) list.push(Math.min(h - 1, Math.floor((k + 0.5) * h / rows)));
  R.render(w, h, scene, {rows: list.filter((_, k) => k % 4 === 0)});
  const t0 = process.hrtime.bigint();
  let r;
  for (let i = 0; i < reps; i++) r = R.render(w, h, scene, {rows: list});
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  emit({cmd, w, h, rows, reps, sample: 'evenly spaced rows', pixels: r.pixels * reps, ms, mpixel_per_s: r.pixels * reps / ms / 1e3, rays: r.rays * reps,
    shadowRays: r.shadowRays * reps, sphereTests: r.sphereTests * reps, node: process.version, threads: 1});
} else {
  console.error('usage: see header of oracle/cli.js');
  process.exit(2);
}
