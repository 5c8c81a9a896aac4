'use strict';
// scene object -> pointer-free blob of include/rt_hip.h (rt_scene_header + tables),
// and scene <-> JSON (the form the Python bench/tests read).

const fs = require('fs');
const path = require('path');
const S = require('./scene.js');

const RT_SCENE_MAGIC = 0x31535452, RT_ABI_VERSION = 2;
const HEADER_BYTES = 208, SPHERE_BYTES = 192, TEXDESC_BYTES = 16;

function flattenScene(scene) {
  S.validateScene(scene);
  const nObj = scene.objects.length, nLight = scene.lights.length, nTex = scene.textures.length;
  const objectsOff = HEADER_BYTES;
  const lightsOff = objectsOff + nObj * SPHERE_BYTES;
  const texOff = lightsOff + nLight * 24;
  let cursor = texOff + nTex * TEXDESC_BYTES;
  const texelOff = scene.textures.map((t) => { const o = cursor; cursor += (t.texels.length + 7) & ~7; return o; });
  const total = cursor;

  const buf = new ArrayBuffer(total);
  const dv = new DataView(buf);
  const u8 = new Uint8Array(buf);
  const LE = true;
  let o = 0;
  const u32 = (v) => { dv.setUint32(o, v, LE); o += 4; };
  const u64 = (v) => { dv.setUint32(o, v >>> 0, LE); dv.setUint32(o + 4, Math.floor(v / 4294967296), LE); o += 8; };
  const f64 = (v) => { dv.setFloat64(o, v, LE); o += 8; };
  const vec = (v, n) => { for (let i = 0; i < n; i++) f64(v[i]); };

  u32(RT_SCENE_MAGIC); u32(RT_ABI_VERSION); u64(total);
  vec(scene.camera.origin, 3); vec(scene.camera.axisX, 3); vec(scene.camera.axisY, 3); vec(scene.camera.axisZ, 3);
  f64(scene.fovDeg); f64(scene.light_intensity); f64(scene.epsilon); vec(scene.miss_color || [1, 0, 0], 3);
  u32(scene.segs); u32(scene.supersample || 1); u32(nObj); u32(nLight); u32(nTex); u32(0);
  u64(objectsOff); u64(lightsOff); u64(texOff);
  if (o !== HEADER_BYTES) throw new Error('flatten: header size drifted');

  for (const obj of scene.objects) {
    const m = obj.mtl, s = m.sampler;
    const start = o;
    vec(obj.origin, 3); f64(obj.r2);
    vec(m.color, 3); f64(m.specular_exponent);
    vec(m.albedo, 5); f64(m.refract_index);
    if (s.kind === S.SAMPLER_CHECKER) { f64(s.freqU); f64(s.freqV); vec(s.colors[0], 3); vec(s.colors[1], 3); }
    else if (s.kind === S.SAMPLER_STARS) { f64(s.threshold); f64(s.scale); for (let i = 0; i < 6; i++) f64(0); }
    else { for (let i = 0; i < 8; i++) f64(0); }
    dv.setInt32(o, s.kind, LE); o += 4;
    dv.setInt32(o, s.kind === S.SAMPLER_TEXTURE ? s.texture : -1, LE); o += 4;
    f64(0);
    if (o - start !== SPHERE_BYTES) throw new Error('flatten: sphere size drifted');
  }
  for (const l of scene.lights) vec(l, 3);
  scene.textures.forEach((t, i) => { u32(t.width); u32(t.height); u64(texelOff[i]); });
  scene.textures.forEach((t, i) => { u8.set(t.texels, texelOff[i]); });
  return buf;
}

// JSON form: numbers round-trip exactly (shortest-repr doubles).  Textures above 4 KiB are
// written beside the JSON as raw .rgba files and referenced by name; small ones inline (base64).
function sceneToJSON(scene, name, dir) {
  const textures = scene.textures.map((t, i) => {
    if (t.texels.length <= 4096) return {width: t.width, height: t.height, base64: Buffer.from(t.texels).toString('base64')};
    const file = (t.name || (name + '_tex' + i)) + '_' + t.width + 'x' + t.height + '.rgba';
    if (dir) fs.writeFileSync(path.join(dir, file), Buffer.from(t.texels));
    return {width: t.width, height: t.height, file};
  });
  return JSON.stringify({
    name, camera: scene.camera, fovDeg: scene.fovDeg, segs: scene.segs, supersample: scene.supersample || 1,
    light_intensity: scene.light_intensity, epsilon: scene.epsilon, lights: scene.lights,
    objects: scene.objects.map((ob) => ({origin: ob.origin, r2: ob.r2, mtl: ob.mtl})),
    textures,
  }, null, 1);
}

function sceneFromJSON(text, dir) {
  const j = JSON.parse(text);
  const textures = j.textures.map((t) => {
    const bytes = t.base64 !== undefined ? Buffer.from(t.base64, 'base64') : fs.readFileSync(path.join(dir, t.file));
    return S.textureFromRGBA(t.width, t.height, bytes);
  });
  return S.createScene({
    camera: j.camera, fovDeg: j.fovDeg, segs: j.segs, supersample: j.supersample, light_intensity: j.light_intensity,
    epsilon: j.epsilon, lights: j.lights, textures,
    objects: j.objects.map((ob) => ({origin: ob.origin, r2: ob.r2, mtl: ob.mtl})),
  });
}

module.exports = {flattenScene, sceneToJSON, sceneFromJSON, HEADER_BYTES, SPHERE_BYTES, TEXDESC_BYTES};
