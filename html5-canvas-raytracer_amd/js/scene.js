'use strict';
// Host-side scene model for render(width, height, scene).
//
// The reference keeps its scene as locals/literals of main()
// (/root/reference/main.js:85-163) and as literals inside intersectWorld
// (:283-284 lights, :194 depth).  This module gives those a home: every field
// below names the reference local it replaces.  Helper names follow the
// reference's constructors (createMaterial :397, createSphere :408,
// createTexture :339, checkerTexture :353, loadTexture :375, lookAt :92) so
// that reference scene code ports by renaming; the implementations are ours.
//
// Sampler closures cannot cross to a GPU, so the reference's `mtl.sampler`
// protocol (:404, :126-157) is ENUMERATED:
//   kind 0 'color'    constant mtl.color                       (:404)
//   kind 1 'texture'  nearest-texel lookup at hit.u,hit.v       (:143-145, :343-351)
//   kind 2 'checker'  sphere checker on its own u,v             (:126-133)
//   kind 3 'stars'    night stars :135-139 with Math.random() replaced by a counter-based hash of (sample index,
//                     position in the ray tree): deterministic, NOT comparable with the random reference

const {readPNG} = require('./png.js');

const SAMPLER_COLOR = 0, SAMPLER_TEXTURE = 1, SAMPLER_CHECKER = 2, SAMPLER_STARS = 3;

function sub(a, b) { return [a[0] - b[0], a[1] - b[1], a[2] - b[2]]; }
function cross(a, b) {
  return [a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]];
}
function dot(a, b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
function unit(v) {                       // multiply by 1/len, as the reference does (main.js:62-66)
  const l = Math.sqrt(dot(v, v));
  if (l === 0) return v;
  const s = 1 / l;
  return [v[0] * s, v[1] * s, v[2] * s];
}

// main.js:92-100 — camera frame from origin/target/up.
function lookAt(org, tgt, up) {
  const z = sub(tgt, org);
  const x = cross(up, z);
  const y = cross(z, x);
  return {origin: [org[0], org[1], org[2]], axisX: unit(x), axisY: unit(y), axisZ: unit(z)};
}

// main.js:397-406
function createMaterial(color, albedo, se, ri) {
  return {
    color: color.slice(0, 3),
    albedo: albedo.slice(0, 5),          // ambient, diffuse, specular, reflect, refract
    specular_exponent: se,
    refract_index: ri,
    sampler: {kind: SAMPLER_COLOR},
  };
}

// main.js:408-418
function createSphere(o, r, m) {
  const r2 = r * r;
  return {origin: o.slice(0, 3), r2: r2, mtl: m, surface_area: 4 * Math.PI * r2};
}

// main.js:339-341
function createTexture() {
  return {width: 0, height: 0, texels: new Uint8Array(0), loaded: false};
}

// main.js:353-373 — texel bytes go through the same clamp/round as a
// Uint8ClampedArray store would apply in the browser.
function checkerTexture(texture, width, height, color1, color2) {
  const t = new Uint8ClampedArray(width * height * 4);
  for (let y = 0; y < height; y++) {
    for (let x = 0; x < width; x++) {
      const c = ((x ^ y) & 1) ? color2 : color1;
      const i = (y * width + x) * 4;
      t[i] = 255 * c[0]; t[i + 1] = 255 * c[1]; t[i + 2] = 255 * c[2]; t[i + 3] = 255;
    }
  }
  texture.width = width; texture.height = height;
  texture.texels = new Uint8Array(t.buffer);
  texture.loaded = true;
  return texture;
}

// main.js:375-395 — synchronous here: the host decodes the PNG itself.
function loadTexture(texture, src) {
  const img = readPNG(src);
  texture.width = img.width; texture.height = img.height;
  texture.texels = img.data; texture.loaded = true;
  return texture;
}

function textureFromRGBA(width, height, rgba) {
  if (rgba.length !== width * height * 4) throw new Error('textureFromRGBA: size mismatch');
  return {width, height, texels: new Uint8Array(rgba), loaded: true};
}

function colorSampler() { return {kind: SAMPLER_COLOR}; }
function textureSampler(textureIndex) { return {kind: SAMPLER_TEXTURE, texture: textureIndex}; }
// main.js:126-133 with its literals (5000, 2500, [[1,1,0],[1,0,1]]) as parameters
function checkerSampler(freqU, freqV, colors) {
  return {kind: SAMPLER_CHECKER, freqU, freqV, colors: [colors[0].slice(0, 3), colors[1].slice(0, 3)]};
}

// main.js:135-139 with its literals (0.001, 1000) as parameters and a hash in place of Math.random()
function starsSampler(threshold, scale) {
  return {kind: SAMPLER_STARS, threshold: threshold === undefined ? 0.001 : threshold, scale: scale === undefined ? 1000 : scale};
}

// main.js:159-163 — ascending surface_area / distance-to-camera; Array.prototype.sort is
// stable in V8 >= 7.0, which is what pins the tie order (SURVEY q4).
function sortObjects(objects, cameraOrigin) {
  const key = (o) => o.surface_area / Math.sqrt(dot(sub(o.origin, cameraOrigin), sub(o.origin, cameraOrigin)));
  return objects.slice().sort((a, b) => key(a) - key(b));
}

function createScene(opts) {
  const scene = {
    camera: opts.camera || lookAt([0, 1.5, 10], [0, 1.5, 0], [0, 1, 0]),      // main.js:85-90
    fovDeg: opts.fovDeg === undefined ? 60 : opts.fovDeg,                       // main.js:102
    segs: opts.segs === undefined ? 8 : opts.segs,                              // main.js:194
    objects: opts.objects || [],
    lights: opts.lights || [[5.0, 10.0, 5.0], [5.0, 10.0, 0.0]],                // main.js:283
    light_intensity: opts.light_intensity === undefined ? 50 : opts.light_intensity, // main.js:284
    textures: opts.textures || [],
    epsilon: opts.epsilon === undefined ? 0.001 : opts.epsilon,                 // main.js:430-436
    supersample: opts.supersample || 1,                                         // 1, or k = 2 (cfg5), 3, 4: k x k box of the kw x kh frame
  };
  validateScene(scene);
  return scene;
}

function isVec(v, n) {
  return (Array.isArray(v) || ArrayBuffer.isView(v)) && v.length === n && Array.prototype.every.call(v, (x) => typeof x === 'number');
}

function validateScene(scene) {
  const c = scene.camera;
  if (!c || !isVec(c.origin, 3) || !isVec(c.axisX, 3) || !isVec(c.axisY, 3) || !isVec(c.axisZ, 3)) throw new Error('scene.camera must hold origin/axisX/axisY/axisZ 3-vectors');
  if (!Number.isInteger(scene.segs) || scene.segs < 0 || scene.segs > 16) throw new Error('scene.segs must be an integer in [0,16]');
  if (![1, 2, 3, 4].includes(scene.supersample)) throw new Error('scene.supersample must be 1, 2, 3 or 4');
  if (!Array.isArray(scene.objects) || scene.objects.length < 1 || scene.objects.length > 256) throw new Error('scene.objects must hold 1..256 spheres');
  if (!Array.isArray(scene.lights) || scene.lights.length > 16) throw new Error('scene.lights must hold 0..16 lights');
  scene.lights.forEach((l) => { if (!isVec(l, 3)) throw new Error('light must be a 3-vector'); });
  scene.textures.forEach((t, i) => {
    if (!(t.width > 0 && t.height > 0) || t.texels.length !== t.width * t.height * 4) throw new Error('texture ' + i + ' is not loaded RGBA8');
  });
  scene.objects.forEach((o, i) => {
    if (!isVec(o.origin, 3) || typeof o.r2 !== 'number') throw new Error('object ' + i + ': origin/r2');
    const m = o.mtl;
    if (!m || !isVec(m.color, 3) || !isVec(m.albedo, 5)) throw new Error('object ' + i + ': material');
    const s = m.sampler;
    if (typeof s === 'function') throw new Error('object ' + i + ': sampler closures cannot cross to the GPU; use colorSampler/textureSampler/checkerSampler');
    if (!s || ![SAMPLER_COLOR, SAMPLER_TEXTURE, SAMPLER_CHECKER, SAMPLER_STARS].includes(s.kind)) throw new Error('object ' + i + ': unsupported sampler kind (use colorSampler/textureSampler/checkerSampler/starsSampler)');
    if (s.kind === SAMPLER_STARS && !(typeof s.threshold === 'number' && typeof s.scale === 'number')) throw new Error('object ' + i + ': stars sampler parameters');
    if (s.kind === SAMPLER_TEXTURE && !(Number.isInteger(s.texture) && s.texture >= 0 && s.texture < scene.textures.length)) throw new Error('object ' + i + ': texture index out of range');
    if (s.kind === SAMPLER_CHECKER && !(typeof s.freqU === 'number' && typeof s.freqV === 'number' && isVec(s.colors[0], 3) && isVec(s.colors[1], 3))) throw new Error('object ' + i + ': checker sampler parameters');
  });
}

module.exports = {
  SAMPLER_COLOR, SAMPLER_TEXTURE, SAMPLER_CHECKER, SAMPLER_STARS, starsSampler,
  lookAt, createMaterial, createSphere, createTexture, checkerTexture, loadTexture, textureFromRGBA,
  colorSampler, textureSampler, checkerSampler, sortObjects, createScene, validateScene,
};
