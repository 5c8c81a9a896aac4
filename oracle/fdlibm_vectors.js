// ORACLE / TEST INFRASTRUCTURE - test vectors for oracle/fdlibm_trig.h from the JS engine itself:  node oracle/fdlibm_vectors.js <out file> [n]
// test vectors from V8: random and special inputs, outputs as hex bit patterns
const N = +(process.argv[3] || 400000);
let s = 12345;
function rnd() { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; }
const buf = new DataView(new ArrayBuffer(8));
const hex = (x) => { buf.setFloat64(0, x); return buf.getUint32(0).toString(16).padStart(8, '0') + buf.getUint32(4).toString(16).padStart(8, '0'); };
const sp = [0, -0, 1, -1, 0.5, -0.5, 1e-300, -1e-300, 1e-10, 0.4375, 0.6875, 1.1875, 2.4375, 0.975, 0.9999999999999999, 1e20, -1e20, Infinity, -Infinity, NaN, 3.8845878744069696e-16, 0.5453115271691488, 1 / 3, 2 / 3];
const out = [];
for (const a of sp) for (const b of sp) out.push(['2', a, b]);
for (const a of sp) out.push(['s', a, 0]);
for (let i = 0; i < N; i++) {
  const k = i % 4;
  let y, x;
  if (k === 0) { y = rnd() * 2 - 1; x = rnd() * 2 - 1; }
  else if (k === 1) { y = (rnd() * 2 - 1) * Math.pow(2, Math.floor(rnd() * 80 - 40)); x = (rnd() * 2 - 1) * Math.pow(2, Math.floor(rnd() * 80 - 40)); }
  else if (k === 2) { y = (rnd() * 2 - 1) * 1e-15; x = -(rnd()); }
  else { y = rnd() * 2 - 1; x = (rnd() * 2 - 1) * 1e-12; }
  out.push(['2', y, x]);
  out.push(['s', rnd() * 2 - 1, 0]);
  if (k === 0) out.push(['s', (rnd() < 0.5 ? -1 : 1) * (1 - rnd() * 1e-3), 0]);
}
const lines = out.map(([t, a, b]) => t === '2' ? `2 ${hex(a)} ${hex(b)} ${hex(Math.atan2(a, b))}` : `s ${hex(a)} ${hex(0)} ${hex(Math.asin(a))}`);
require('fs').writeFileSync(process.argv[2], lines.join('\n') + '\n');
console.log(lines.length, process.version);
