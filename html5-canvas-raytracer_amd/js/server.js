'use strict';
// Tiny HTTP bridge (Node core modules only) that puts the MI355X path back behind a browser canvas
// (SURVEY.md §8(f)-2).  The page it serves is the reference's page shell in spirit (index.html:15:
// one full-window <canvas id='canvasID'>); instead of running main.js's scanline loop
// (main.js:180-201) in the browser it asks this host for the frame and does the one thing the
// browser is needed for: context.putImageData.  The elapsed-ms overlay of main.js:204-210 is kept.
//
//   GET /                               the page
//   GET /frame?scene=h8&w=1280&h=720    raw RGBA8, w*h*4 bytes (ImageData.data layout);
//                                       headers X-Width, X-Height, X-Kernel-Ms, X-Total-Ms, X-Build, X-Report ('build #741.r4 (12ms)')
//   GET /frame?...&progressive=8        the same bytes as a CHUNKED response, one chunk per row band as it leaves the
//                                       GPU (renderProgressive): the page paints top to bottom like the reference's
//                                       scanline loop (main.js:183-201)
//   GET /scenes                         JSON list of scene names
// Errors (no GPU, bad scene, bad size) are JSON with status 4xx/5xx; never a CPU-rendered frame.
//
//   node html5-canvas-raytracer_amd/js/server.js [port]

const http = require('http');
const fs = require('fs');
const path = require('path');
const url = require('url');
const RT = require('./index.js');
const F = require('./flatten.js');

const SCENES_DIR = path.join(__dirname, '..', 'scenes');

const PAGE = `<!DOCTYPE html>
<html><head><meta charset='utf-8'><title>mi355x-sphere-tracer</title>
<style>html,body{margin:0;width:100%;height:100%;overflow:hidden;background:#000}canvas{display:block}</style></head>
<body><canvas id='canvasID'></canvas><script>
(async function () {
  const canvas = document.getElementById('canvasID');
  canvas.width = document.body.clientWidth; canvas.height = document.body.clientHeight;
  const ctx = canvas.getContext('2d');
  const scene = new URLSearchParams(location.search).get('scene') || 'default14_stars';
  const t0 = Date.now();
  const w = canvas.width, h = canvas.height;
  const r = await fetch('/frame?scene=' + scene + '&w=' + w + '&h=' + h + '&progressive=8');
  if (!r.ok) { ctx.fillStyle = '#f44'; ctx.font = '16px monospace'; ctx.fillText((await r.json()).error, 8, 24); return; }
  // paint whole rows as the chunks arrive (the reference paints one scanline per macrotask)
  const data = new Uint8ClampedArray(w * h * 4);
  const reader = r.body.getReader();
  let got = 0, painted = 0;
  for (;;) {
    const {done, value} = await reader.read();
    if (done) break;
    data.set(value, got); got += value.length;
    const rows = Math.floor(got / (w * 4));
    if (rows > painted) { ctx.putImageData(new ImageData(data.subarray(painted * w * 4, rows * w * 4), w, rows - painted), 0, painted); painted = rows; }
  }
  // the reference's end-of-frame overlay (main.js:203-210), same string, font and placement: 'build #<id> (<elapsed>ms)' with
  // elapsed measured here, in the browser, around the whole frame as the reference measures it
  const message = 'build #' + (r.headers.get('X-Build') || '?') + ' (' + (Date.now() - t0) + 'ms)';
  ctx.font = '16px monospace'; ctx.textAlign = 'left'; ctx.textBaseline = 'top'; ctx.fillStyle = '#ffffff';
  ctx.fillText(message, 0, 0);
})();
</script></body></html>`;

const sceneCache = new Map();
function loadNamedScene(name) {
  if (!/^[A-Za-z0-9_]+$/.test(name)) throw Object.assign(new Error('bad scene name'), {status: 400});
  if (!sceneCache.has(name)) {
    const p = path.join(SCENES_DIR, name + '.json');
    if (!fs.existsSync(p)) throw Object.assign(new Error('unknown scene ' + name), {status: 404});
    sceneCache.set(name, F.sceneFromJSON(fs.readFileSync(p, 'utf8'), SCENES_DIR));
  }
  return sceneCache.get(name);
}

function listScenes() {
  return fs.readdirSync(SCENES_DIR).filter((f) => f.endsWith('.json')).map((f) => f.slice(0, -5)).sort();
}

function sendJSON(res, status, obj) {
  const body = JSON.stringify(obj);
  res.writeHead(status, {'Content-Type': 'application/json', 'Content-Length': Buffer.byteLength(body)});
  res.end(body);
}

function createServer(opts) {
  opts = opts || {};
  const maxPixels = opts.maxPixels || 16384 * 16384;
  return http.createServer((req, res) => {
    const u = url.parse(req.url, true);
    if (req.method !== 'GET') return sendJSON(res, 405, {error: 'GET only'});
    if (u.pathname === '/') {
      res.writeHead(200, {'Content-Type': 'text/html; charset=utf-8', 'Content-Length': Buffer.byteLength(PAGE)});
      return res.end(PAGE);
    }
    if (u.pathname === '/scenes') return sendJSON(res, 200, {scenes: listScenes()});
    if (u.pathname === '/frame') {
      const w = parseInt(u.query.w, 10), h = parseInt(u.query.h, 10);
      if (!(w > 0 && h > 0 && w <= 65536 && h <= 65536 && w * h <= maxPixels)) return sendJSON(res, 400, {error: 'w and h must be positive integers within the frame limit'});
      let scene;
      try { scene = loadNamedScene(String(u.query.scene || 'default14_stars')); } catch (e) { return sendJSON(res, e.status || 500, {error: e.message}); }
      const bands = parseInt(u.query.progressive || '0', 10);
      if (bands > 0) {
        // chunked: each band is written as soon as it is in the pinned frame; headers cannot carry the timings any more
        // (they are known at the end), so they travel as HTTP trailers
        let started = false;
        return RT.renderProgressive(w, h, scene, {bands: Math.min(bands, 64), onBand: (b) => {
          if (!started) {
            res.writeHead(200, {'Content-Type': 'application/octet-stream', 'X-Width': w, 'X-Height': h, 'X-Build': RT.buildId(), 'Cache-Control': 'no-store',
              'Trailer': 'X-Kernel-Ms, X-Total-Ms, X-Report'});
            started = true;
          }
          res.write(Buffer.from(b.data.buffer, b.data.byteOffset, b.data.length));
        }}).then((data) => {
          res.addTrailers({'X-Kernel-Ms': data.stats.kernel_ms.toFixed(3), 'X-Total-Ms': data.stats.total_ms.toFixed(3), 'X-Report': data.stats.report});
          res.end();
        }).catch((e) => { if (started) res.destroy(e); else sendJSON(res, 503, {error: e.message}); });
      }
      // renderAsync keeps the event loop free while the GPU works; the reply streams the pinned frame
      return RT.renderAsync(w, h, scene).then((data) => {
        res.writeHead(200, {'Content-Type': 'application/octet-stream', 'Content-Length': data.length, 'X-Width': w, 'X-Height': h,
          'X-Kernel-Ms': data.stats.kernel_ms.toFixed(3), 'X-Total-Ms': data.stats.total_ms.toFixed(3), 'X-Build': data.stats.build, 'X-Report': data.stats.report,
          'Cache-Control': 'no-store'});
        res.end(Buffer.from(data.buffer, data.byteOffset, data.length));
      }).catch((e) => sendJSON(res, 503, {error: e.message}));
    }
    return sendJSON(res, 404, {error: 'not found'});
  });
}

module.exports = {createServer, listScenes, PAGE};

if (require.main === module) {
  const port = parseInt(process.argv[2] || '8080', 10);
  createServer().listen(port, '127.0.0.1', () => console.log('mi355x-sphere-tracer bridge on http://127.0.0.1:' + port + '/?scene=default14_stars'));
}
