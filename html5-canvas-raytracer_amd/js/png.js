'use strict';
// Synchronous PNG -> RGBA8 decoder (zlib only, no DOM).
//
// Replaces the browser path of the reference's loadTexture
// (/root/reference/main.js:375-395: Image -> offscreen canvas -> getImageData),
// which cannot run under Node.  Output layout is what getImageData returns:
// row-major, top row first, 4 bytes per texel R,G,B,A (straight alpha).
//
// Supported: 8-bit and 16-bit (reduced to 8) colour types 0/2/3/4/6, and
// sub-byte depths 1/2/4 for grey and palette, non-interlaced.  tRNS for palette
// images.  gAMA/iCCP are ignored (browsers ignore them too when absent, and the
// reference's two textures carry neither).

const zlib = require('zlib');
const fs = require('fs');

const SIG = Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]);

function paeth(a, b, c) {
  const p = a + b - c;
  const pa = Math.abs(p - a), pb = Math.abs(p - b), pc = Math.abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  if (pb <= pc) return b;
  return c;
}

function decodePNG(buf) {
  if (buf.length < 8 || buf.compare(SIG, 0, 8, 0, 8) !== 0) throw new Error('png: bad signature');
  let off = 8;
  let width = 0, height = 0, depth = 0, ctype = 0, interlace = 0;
  let plte = null, trns = null;
  const idat = [];
  let sawIHDR = false, sawIEND = false;
  while (off + 8 <= buf.length) {
    const len = buf.readUInt32BE(off);
    const type = buf.toString('latin1', off + 4, off + 8);
    const body = buf.slice(off + 8, off + 8 + len);
    if (body.length !== len) throw new Error('png: truncated chunk ' + type);
    off += 12 + len;
    if (type === 'IHDR') {
      width = body.readUInt32BE(0); height = body.readUInt32BE(4);
      depth = body[8]; ctype = body[9]; interlace = body[12];
      sawIHDR = true;
    } else if (type === 'PLTE') plte = body;
    else if (type === 'tRNS') trns = body;
    else if (type === 'IDAT') idat.push(body);
    else if (type === 'IEND') { sawIEND = true; break; }
  }
  if (!sawIHDR || !sawIEND) throw new Error('png: missing IHDR/IEND');
  if (interlace !== 0) throw new Error('png: interlaced images are not supported');
  const channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype];
  if (channels === undefined) throw new Error('png: bad colour type ' + ctype);
  if (![1, 2, 4, 8, 16].includes(depth)) throw new Error('png: bad bit depth ' + depth);
  if (depth < 8 && !(ctype === 0 || ctype === 3)) throw new Error('png: bad depth for colour type');
  if (ctype === 3 && !plte) throw new Error('png: palette image without PLTE');

  const raw = zlib.inflateSync(Buffer.concat(idat));
  const bpp = Math.max(1, (channels * depth) >> 3);       // filter unit in bytes
  const stride = (width * channels * depth + 7) >> 3;     // bytes per scanline
  if (raw.length < (stride + 1) * height) throw new Error('png: short image data');

  // un-filter in place into `pix`
  const pix = Buffer.alloc(stride * height);
  for (let y = 0; y < height; y++) {
    const ft = raw[y * (stride + 1)];
    const src = y * (stride + 1) + 1;
    const dst = y * stride;
    const up = dst - stride;
    for (let x = 0; x < stride; x++) {
      const a = x >= bpp ? pix[dst + x - bpp] : 0;
      const b = y > 0 ? pix[up + x] : 0;
      const c = (x >= bpp && y > 0) ? pix[up + x - bpp] : 0;
      let v = raw[src + x];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: throw new Error('png: bad filter ' + ft);
      }
      pix[dst + x] = v & 255;
    }
  }

  const out = new Uint8Array(width * height * 4);
  const sample = (row, i) => {         // i-th sample of a scanline, scaled to 8 bits for grey
    if (depth === 8) return pix[row + i];
    if (depth === 16) return pix[row + 2 * i];
    const per = 8 / depth;
    const byte = pix[row + Math.floor(i / per)];
    const shift = (per - 1 - (i % per)) * depth;
    return (byte >> shift) & ((1 << depth) - 1);
  };
  const greyScale = depth < 8 ? 255 / ((1 << depth) - 1) : 1;
  for (let y = 0; y < height; y++) {
    const row = y * stride;
    for (let x = 0; x < width; x++) {
      const o = (y * width + x) * 4;
      if (ctype === 3) {
        const idx = sample(row, x);
        if (idx * 3 + 2 >= plte.length) throw new Error('png: palette index out of range');
        out[o] = plte[idx * 3]; out[o + 1] = plte[idx * 3 + 1]; out[o + 2] = plte[idx * 3 + 2];
        out[o + 3] = (trns && idx < trns.length) ? trns[idx] : 255;
      } else if (ctype === 0) {
        const g = Math.round(sample(row, x) * greyScale);
        out[o] = g; out[o + 1] = g; out[o + 2] = g; out[o + 3] = 255;
      } else if (ctype === 4) {
        const g = sample(row, 2 * x);
        out[o] = g; out[o + 1] = g; out[o + 2] = g; out[o + 3] = sample(row, 2 * x + 1);
      } else if (ctype === 2) {
        out[o] = sample(row, 3 * x); out[o + 1] = sample(row, 3 * x + 1);
        out[o + 2] = sample(row, 3 * x + 2); out[o + 3] = 255;
      } else {
        out[o] = sample(row, 4 * x); out[o + 1] = sample(row, 4 * x + 1);
        out[o + 2] = sample(row, 4 * x + 2); out[o + 3] = sample(row, 4 * x + 3);
      }
    }
  }
  return {width, height, data: out};
}

function readPNG(path) {
  return decodePNG(fs.readFileSync(path));
}

module.exports = {decodePNG, readPNG};
