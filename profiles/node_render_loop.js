'use strict';
// What JavaScript waits for: render(width, height, scene) from Node through the N-API layer, end to end (scene flattening, the
// pinned ImageData buffer, the launch, the pixels in host memory, the typed array over them) - the replacement of the reference's
// redraw() + ImageData hand-over (main.js:83, 180-201).   node --expose-gc profiles/node_render_loop.js [scene] [w] [h] [calls]
//   "into":      the frame of the first render is filled again by every later one (the reference's own pattern: ImageData created
//                once, main.js:83) - nothing is allocated per frame
//   "fresh":     a new frame per call; the old ones go back to the library's pinned pool when the collector finalises them
//                (an event-loop turn and a collection between calls, outside the timed window)
//   "pageable":  into a plain Uint8ClampedArray (what a canvas ImageData.data is)
const path = require('path');
const fs = require('fs');
const rt = require(path.join(__dirname, '..', 'html5-canvas-raytracer_amd', 'js', 'index.js'));
const F = require(path.join(__dirname, '..', 'html5-canvas-raytracer_amd', 'js', 'flatten.js'));
const SC = path.join(__dirname, '..', 'html5-canvas-raytracer_amd', 'scenes');
const name = process.argv[2] || 'h8', w = +(process.argv[3] || 3840), h = +(process.argv[4] || 2160), calls = +(process.argv[5] || 60);
const scene = F.sceneFromJSON(fs.readFileSync(path.join(SC, name + '.json'), 'utf8'), SC);
const med = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];
const ms = (t0) => Number(process.hrtime.bigint() - t0) / 1e6;
const turn = () => new Promise((r) => setImmediate(r));

(async () => {
  rt.init(1);
  const out = {scene: name, width: w, height: h, calls, frame_bytes: w * h * 4, build: rt.buildId(), modes: {}};
  const first = rt.render(w, h, scene);
  const ref = Buffer.from(first.buffer, first.byteOffset, first.length);
  const refSum = ref.reduce((a, b) => (a + b) >>> 0, 0);
  { const f = []; for (let i = 0; i < 20; i++) { const t0 = process.hrtime.bigint(); rt.flattenScene(scene); f.push(ms(t0)); } out.flatten_scene_ms_median = +med(f).toFixed(4); }
  for (const mode of ['into', 'fresh', 'pageable']) {
    const wall = [], total = [], kernel = [];
    let frame = mode === 'pageable' ? new Uint8ClampedArray(w * h * 4) : first, same = true;
    for (let i = 0; i < calls + 5; i++) {
      const t0 = process.hrtime.bigint();
      const data = mode === 'fresh' ? rt.render(w, h, scene) : rt.render(w, h, scene, {into: frame});
      const t = ms(t0);
      if (i >= 5) { wall.push(t); total.push(data.stats.total_ms); kernel.push(data.stats.kernel_ms); }
      if (i === calls + 4) same = Buffer.compare(Buffer.from(data.buffer, data.byteOffset, data.length), ref) === 0;
      if (mode === 'fresh') { await turn(); if (global.gc) global.gc(); await turn(); }
    }
    out.modes[mode] = {render_wall_ms_median: +med(wall).toFixed(4), render_wall_ms_min: +Math.min(...wall).toFixed(4), rt_render_total_ms_median: +med(total).toFixed(4),
      kernel_ms_median: +med(kernel).toFixed(4), mpixel_per_s_end_to_end: +(w * h / med(wall) / 1e3).toFixed(1), host_GBs_end_to_end: +(w * h * 4 / med(wall) / 1e6).toFixed(2),
      same_bytes_as_the_first_frame: same};
  }
  out.checksum = refSum;
  console.log(JSON.stringify(out));
  rt.shutdown();
})().catch((e) => { console.error(e); process.exit(1); });
