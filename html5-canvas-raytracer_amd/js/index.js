'use strict';
// render(width, height, scene) — the drop-in entry point for the reference's per-pixel path
// (redraw()/spanish() + intersectWorld, /root/reference/main.js:180-201, :216-451), backed by the
// MI355X HIP kernel through the N-API shim.  The returned Uint8ClampedArray has the layout of
// ImageData.data (main.js:83, :195-198), so it drops in behind the canvas:
//     ctx.putImageData(new ImageData(render(w, h, scene), w, h), 0, 0)
// There is NO JavaScript rendering fallback here: if the addon or the GPU is missing, this throws.

const path = require('path');
const {flattenScene} = require('./flatten.js');
const scene = require('./scene.js');
const scenes = require('./scenes.js');

let addon = null;
function native() {
  if (addon) return addon;
  const p = path.join(__dirname, '..', 'napi', 'rt_napi.node');
  try { addon = require(p); } catch (e) {
    throw new Error('html5-canvas-raytracer_amd: native addon ' + p + ' is not built or cannot load (' + e.message +
      '); run `python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback for render().');
  }
  return addon;
}

const FLAG_COUNT = 1, FLAG_STRICT_FP = 2;
let inited = false;
// maxDevices: GPUs rt_render may shard one frame over (RCCL gather); default 1.  Pass 0 for every visible GPU.
function init(maxDevices) { const n = native().init(maxDevices === undefined ? 1 : maxDevices); inited = true; return n; }
function flagsOf(opts) { return ((opts && opts.count) ? FLAG_COUNT : 0) | ((opts && opts.strictFp) ? FLAG_STRICT_FP : 0); }

// -> Uint8ClampedArray of length 4*width*height over a pinned host buffer; `.stats` carries timings.
// opts.into: the frame to fill instead of a new one - the reference creates its ImageData once (main.js:83) and every redraw
// writes into it again (main.js:195-200).  Hand in what an earlier render() returned (pinned memory: the GPU stores into it
// directly, nothing is allocated) - or any Uint8ClampedArray of 4*width*height bytes, e.g. a canvas ImageData.data (pageable
// memory: the frame is copied out of the GPU's memory, about half the rate).  Returns that same array.
function render(width, height, sceneObj, opts) {
  if (!inited) init(opts && opts.maxDevices);
  const r = native().render(flattenScene(sceneObj), width, height, flagsOf(opts), (opts && opts.into) || undefined);
  r.data.stats = r.stats;
  return r.data;
}

// Promise variant: the launch and the copy-out run off the event loop (napi_async_work)
function renderAsync(width, height, sceneObj, opts) {
  try { if (!inited) init(opts && opts.maxDevices); } catch (e) { return Promise.reject(e); }
  return native().renderAsync(flattenScene(sceneObj), width, height, flagsOf(opts)).then((r) => { r.data.stats = r.stats; return r.data; });
}

// Progressive variant: the reference shows its frame row by row (one spanish(y) per macrotask, main.js:183-201); here
// the frame is rendered as `bands` row bands (default 8, at most 64) and onBand({firstRow, rows, data}) fires on the main
// thread as each band lands in the frame buffer - `data` is the view of just those rows, `frame` the whole buffer that
// is being filled (usable with putImageData(..., dirtyY) while it fills).  Resolves to the full frame.
function renderProgressive(width, height, sceneObj, opts) {
  try { if (!inited) init(opts && opts.maxDevices); } catch (e) { return Promise.reject(e); }
  const bands = (opts && opts.bands) || 8;
  const onBand = (opts && opts.onBand) || (() => {});
  let r;
  try {
    r = native().renderProgressive(flattenScene(sceneObj), width, height, flagsOf(opts), bands,
      (firstRow, rows) => onBand({firstRow, rows, frame: r.data, data: r.data.subarray(firstRow * width * 4, (firstRow + rows) * width * 4)}));
  } catch (e) { return Promise.reject(e); }
  return r.promise.then((stats) => { r.data.stats = stats; return r.data; });
}

// `const build = '741'` (main.js:3) + this library's revision; every render's `.stats` also carries `.build` and `.report`, the
// reference's end-of-frame string 'build #<id> (<elapsed>ms)' (main.js:204-205) for that render.
function buildId() { return native().buildId(); }

function shutdown() { if (addon) addon.shutdown(); inited = false; }

module.exports = Object.assign({render, renderAsync, renderProgressive, init, shutdown, buildId, flattenScene, scenes, native}, scene);
