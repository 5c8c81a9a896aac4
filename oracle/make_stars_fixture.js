'use strict';
// ORACLE / TEST INFRASTRUCTURE — build container only (needs /root/reference):  node oracle/make_stars_fixture.js
//
// The reference's skybox sampler is random (main.js:135-139: c = Math.random(); c >= 0.001 -> black, else the grey c * 1000), so its
// stars cannot be pinned pixel for pixel (SURVEY 0.5, 8(f)-3).  What CAN be pinned is their statistics.  This script runs the
// UNMODIFIED reference main() with its real Math.random RUNS times at W x H, compares every run with the pinned-random frame (black
// sky) and writes tests/golden/stars_statistics.json:
//   sky_pixels      pixels of the frame that show the sky directly (pure black in the pinned frame and never anything but black or
//                   a grey r = g = b in any run)
//   stars_per_run   how many of them are lit in each run (each is a Bernoulli(0.001) draw per sky pixel)
//   grey_histogram  the lit pixels' byte values over all runs, 16 bins of 16 (the reference's grey is uniform on [0, 1): 255 * c * 1000)
//   changed_elsewhere_per_run   pixels that differ from the pinned frame anywhere else (stars mirrored in reflective spheres)
// The product's deterministic stand-in (a counter-based hash per sample and ray-tree node, 8(f)-3) is held to these numbers by
// tests/test_gpu_parity.py::test_hashed_stars_match_the_references_statistics and, for the restatements, tests/test_oracle.py.
const fs = require('fs');
const path = require('path');
const H = require('./ref_harness.js');
if (!H.available()) { console.error('reference not present at ' + H.REF_DIR); process.exit(2); }
const W = 640, Hh = 360, RUNS = 24;
const pinned = H.runMain(W, Hh);
const n = W * Hh;
const black = new Uint8Array(n), ok = new Uint8Array(n).fill(1);
for (let i = 0; i < n; i++) black[i] = (pinned[4 * i] | pinned[4 * i + 1] | pinned[4 * i + 2]) === 0 ? 1 : 0;
const runs = [];
for (let r = 0; r < RUNS; r++) {
  const f = Buffer.from(H.runMain(W, Hh, {realRandom: true}));
  runs.push(f);
  for (let i = 0; i < n; i++) if (black[i] && !(f[4 * i] === f[4 * i + 1] && f[4 * i] === f[4 * i + 2])) ok[i] = 0;
}
// direct sky: black when pinned, grey-or-black in every run, and not inside the picture's non-sky part.  A reflective sphere that
// mirrors black sky is black too, but its stars are dimmed by the albedo and tinted by the surface colour; to keep the fixture
// about the sampler itself, the sky is taken ABOVE the horizon row only where whole rows are black in the pinned frame.
let skyRows = 0;
for (let y = 0; y < Hh; y++) { let all = true; for (let x = 0; x < W; x++) if (!black[y * W + x]) { all = false; break; } if (all) skyRows++; else break; }
const sky = new Uint8Array(n);
let skyPixels = 0;
for (let i = 0; i < skyRows * W; i++) if (ok[i]) { sky[i] = 1; skyPixels++; }
const hist = new Array(16).fill(0), stars = [], elsewhere = [];
for (const f of runs) {
  let s = 0, e = 0;
  for (let i = 0; i < n; i++) {
    const lit = f[4 * i] !== pinned[4 * i] || f[4 * i + 1] !== pinned[4 * i + 1] || f[4 * i + 2] !== pinned[4 * i + 2];
    if (sky[i]) { if (f[4 * i] > 0) { s++; hist[f[4 * i] >> 4]++; } } else if (lit) e++;
  }
  stars.push(s); elsewhere.push(e);
}
const out = {generator: 'oracle/make_stars_fixture.js', reference: 'termuxinator/html5-canvas-raytracer build #741 main() with its own Math.random (main.js:135-139)',
  node: process.version, w: W, h: Hh, runs: RUNS, sky_rows: skyRows, sky_pixels: skyPixels, probability_in_the_source: 0.001,
  stars_per_run: stars, grey_histogram: hist, changed_elsewhere_per_run: elsewhere,
  note: 'a star whose grey rounds to byte 0 (c * 1000 * 255 < 0.5: 0.2 % of the stars) is not counted: the same rule on both sides'};
fs.writeFileSync(path.join(__dirname, '..', 'tests', 'golden', 'stars_statistics.json'), JSON.stringify(out, null, 1));
console.log(JSON.stringify(out));
