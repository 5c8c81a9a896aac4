'use strict';
// Named synthetic scenes of BASELINE.json `configs` / SURVEY.md §8(d).
// All use the reference's default camera (main.js:85-90) and draw their spheres
// from the reference's object table (main.js:107-124); each is sorted with the
// reference's rule (main.js:159-163) before it is handed to render().

const S = require('./scene.js');

const CAMERA = () => S.lookAt([0, 1.5, 10], [0, 1.5, 0], [0, 1, 0]);

// main.js:108-123, by name
function home()   { const m = S.createMaterial([1, 1, 1], [0, 0.5, 0.8, 0, 0], 10, 1.0); m.sampler = S.checkerSampler(5000, 2500, [[1, 1, 0], [1, 0, 1]]); return S.createSphere([0, -500, 0], 500, m); }
function skybox() { return S.createSphere([0, 0, 0], 5000, S.createMaterial([0, 0, 0], [1, 0, 0, 0, 0], 0, 1.0)); }  // stars with Math.random()>=0.001 == constant black
function earth(tex) { const m = S.createMaterial([1, 1, 1], [1, 0, 0, 0, 0], 0, 1.0); m.sampler = S.textureSampler(tex); return S.createSphere([50, 20, -100], 4.0, m); }
function mars(tex)  { const m = S.createMaterial([1, 1, 1], [1, 0, 0, 0, 0], 0, 1.0); m.sampler = S.textureSampler(tex); return S.createSphere([-50, 20, -100], 2.0, m); }
function matte(tex) { const m = S.createMaterial([1, 1, 1], [0, 1.0, 0.1, 0, 0], 10, 1.0); m.sampler = S.textureSampler(tex); return S.createSphere([0, 0.25, 3], 0.25, m); }
function glass()   { return S.createSphere([-2.5, 0.5, 3], 0.5, S.createMaterial([1, 1, 1], [0, 0.5, 0.2, 0, 0.8], 50, 1.5)); }
function chromeA() { return S.createSphere([1.5, 2.5, 0], 0.5, S.createMaterial([0.5, 0.5, 0.5], [0, 0.5, 1.0, 0.4, 0], 20, 1.0)); }
function chromeB() { return S.createSphere([1.0, 0.25, 3], 0.25, S.createMaterial([0.5, 0.5, 0.5], [0, 0.5, 1.0, 0.5, 0], 20, 1.0)); }
function bubble()  { return S.createSphere([2.5, 0.5, 3], 0.5, S.createMaterial([0.5, 0.5, 0.5], [0, 0.4, 0.5, 0.2, 0.8], 20, 1.0)); }
function mirror()  { return S.createSphere([0, 2.5, -2], 0.5, S.createMaterial([1, 1, 1], [0, 0.1, 0.5, 0.6, 0], 500, 1.0)); }
function metal()   { return S.createSphere([-1.5, 2.5, 0], 0.5, S.createMaterial([1, 1, 1], [0, 0.8, 0.2, 0.1, 0], 50, 1.0)); }
function ornament(x, z, color) { return S.createSphere([x, 1.0, z], 1.0, S.createMaterial(color, [0, 0.8, 0.3, 0.5, 0], 50, 1.0)); }

function finish(objects, textures, opts) {
  const camera = CAMERA();
  return S.createScene(Object.assign({camera, objects: S.sortObjects(objects, camera.origin), textures}, opts));
}

// cfg1: 256x256, 2 untextured spheres, 1 light, depth 1 (misses stay red, main.js:231)
function cfg1() {
  const ground = S.createSphere([0, -500, 0], 500, S.createMaterial([1, 1, 1], [0, 0.5, 0.8, 0, 0], 10, 1.0));
  return finish([ornament(-1.5, 0, [1, 0, 0]), ground], [], {segs: 1, lights: [[5.0, 10.0, 5.0]]});
}

// cfg2: earth + mars textured, checker ground, black sky, depth 2, 2 lights
function cfg2(tex) {
  return finish([home(), skybox(), earth(0), mars(1)], [tex.earth, tex.mars], {segs: 2});
}

// cfg3 "H8": 8 spheres, 2 lights, depth 3 (headline)
function h8(tex, segs) {
  const objs = [home(), skybox(), earth(0), mars(1), mirror(),
    ornament(-1.5, 0, [1, 0, 0]), ornament(1.5, 0, [0, 1, 0]), ornament(0, -2, [0, 0, 1])];
  return finish(objs, [tex.earth, tex.mars], {segs: segs === undefined ? 3 : segs});
}

// The reference's own 14-sphere scene (main.js:107-157) with the stars sampler
// pinned to black (Math.random stubbed to 0.5); depth 8 (main.js:194).
function default14(tex, stars) {
  const checker = S.checkerTexture(S.createTexture(), 16, 8, [0, 0, 0], [1, 1, 1]);
  const sky = skybox();
  if (stars) sky.mtl.sampler = S.starsSampler(0.001, 1000);   // main.js:135-139, hashed instead of Math.random()
  const objs = [home(), sky, earth(0), mars(1), matte(2), glass(), chromeA(), chromeB(), bubble(),
    mirror(), metal(), ornament(-1.5, 0, [1, 0, 0]), ornament(1.5, 0, [0, 1, 0]), ornament(0, -2, [0, 0, 1])];
  return finish(objs, [tex.earth, tex.mars, checker], {segs: 8});
}

// cfg5: 64 spheres from a 32-bit LCG (seed 1), ground + black sky + 62 on a jittered 8x8
// grid (two grid cells left empty), reflective mix, no refraction, depth 5, 2x2 supersample.
function lcg64(segs, supersample) {
  let s = 1;
  const rnd = () => { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
  const objs = [home(), skybox()];
  for (let k = 0; k < 64 && objs.length < 64; k++) {
    const gx = k % 8, gz = (k / 8) | 0;
    if (k === 27 || k === 36) { rnd(); rnd(); rnd(); rnd(); rnd(); rnd(); rnd(); continue; }
    const r = 0.2 + 0.4 * rnd();
    const x = (gx - 3.5) * 1.6 + (rnd() - 0.5) * 0.6;
    const z = 2.0 - gz * 1.6 + (rnd() - 0.5) * 0.6;
    const color = [rnd(), rnd(), rnd()];
    const refl = [0, 0.3, 0.6][Math.floor(rnd() * 3)];
    const se = [10, 20, 50][Math.floor(rnd() * 3)];
    objs.push(S.createSphere([x, r, z], r, S.createMaterial(color, [0, 0.8, 0.3, refl, 0], se, 1.0)));
  }
  return finish(objs, [], {segs: segs === undefined ? 5 : segs, supersample: supersample === undefined ? 2 : supersample});
}

module.exports = {cfg1, cfg2, h8, default14, lcg64};
