// rt_block.h — what the product kernel's launch table states about ONE block of samples, written once for the host
// (rt_tables.cpp: build_launch_table, the CPU tests' oracle and the no-GPU probe) and for the device (rt_tables_gpu.hip: the
// table as the library builds it, one work-item per block).  Plain binary64, IEEE sqrt and division, no contraction (both
// translation units are compiled without it): the two builds give the same words.
//
// A block is the 32 x 8 pixels (x supersample) of one workgroup of the trace kernel.  The rays of its samples lie inside the
// circular cone around the box's centre direction that contains its four corner directions (half-angle alpha); a sphere with
// centre C (seen from the camera) and radius R is met by SOME ray of that cone only if the angle between the cone's axis and C
// is at most alpha + asin(R / |C|).  From that, with margins far above the kernel's rounding:
//   * touched    - can any sphere but the enclosing one show in the block?  If not, and the scene's background is a constant,
//                  the block is SKY: its workgroup stores the constant and traces nothing;
//   * candidates - the (at most two) loop-order spheres the block's primary rays can meet at all (count << 16 | second << 8 |
//                  first), or 0 = no statement: the kernel then tests those and skips its cull;
//   * shadow masks (at most two lights) - where can the block's primary hits lie?  On candidate sphere i at distances
//                  t(c) = c - sqrt(c^2 - k), c = d.C, which over the cone is the interval between t at the largest and at the
//                  smallest c, so inside the ball around the cone's axis point at the mid distance with radius^2 = (dt/2)^2 +
//                  2 t2 m (1 - cos alpha); sphere j can stand between such a point and light k only if the angle between
//                  (Q - L) and (C_j - L) is at most asin(rho / |Q - L|) + asin(R_j / |C_j - L|) and its nearest point is not
//                  beyond the patch.  The union over the block's candidates, per light, is a 16-bit set of loop-order sphere
//                  indices (0xffffffff = no statement: scan everything); with more than 16 loop spheres a light's set is stored
//                  as empty (0) or not (0xffff);
//   * cost       - 1 + the weights of the spheres whose screen rectangle (the primary-ray cull's) touches the block, + what its
//                  mirrors show of the scene's dearest spheres (rt_bounce_cost): what ranks the blocks dearest first.
#ifndef RT_BLOCK_H
#define RT_BLOCK_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

#define RT_TABLE_SKY 1u          /* mark sky blocks */
#define RT_TABLE_MASKS 2u        /* shadow masks */
#define RT_TABLE_CANDS 4u        /* primary candidates */
#define RT_TABLE_WIDE 8u         /* more than 16 loop spheres: a light's set is empty (0) or not (0xffff) */
#define RT_TABLE_RANK 16u        /* list the blocks dearest first */
#define RT_TABLE_GEOMETRY 32u    /* the camera is usable for the cone test (finite, non-zero axis sums, positive projection distance) */
#define RT_TABLE_NO_SKY 64u      /* RT_FLAG_NO_SKY: the table holds no entry for sky blocks (their slots stay zero: no workgroup renders them) */
#define RT_TABLE_SKY_ONLY 128u   /* RT_FLAG_SKY_ONLY: the table holds ONLY the sky runs */
#define RT_TABLE_BOUNCE 256u     /* ranked launch of a scene with a sphere that both reflects and refracts: a block's cost also counts such spheres seen in its mirrors */
#define RT_SKY_RUN_MAX 32u       /* consecutive sky blocks of a row block that share ONE entry */
#define RT_COST_MAX 1023u        /* costs are clamped here (the ranking only has to order the blocks roughly) */

// a sphere as the cone test sees it (one per sphere but the enclosing one, scene order)
struct rt_ball {
  double c[3];                   // unit vector from the camera to the centre (the raw difference when `everywhere`)
  double o[3];                   // the centre
  double len, R, k;              // |centre - camera|, radius x (1 + 1e-7), len^2 - r^2 (with the TRUE radius: the tangent length bounds the hit distances)
  double sin_b, cos_b;           // of asin(R / len)
  double tangent;                // sqrt(max(k, 0)): the distance at which a ray from the camera grazes the sphere
  // as an OCCLUDER seen from light k (at most two lights matter for the masks): centre - light, its length wl, wl - R, and sin / cos of
  // asin(R / wl); `always`: the light is inside the sphere (or nothing finite can be said): it is in every set
  double lw[2][3], wl[2], wl_minus_R[2], s2[2], c2[2];
  uint32_t always[2];
  uint32_t loop;                 // index in the product kernel's loop order (enclosing sphere last)
  uint32_t everywhere;           // the camera is inside / on / too near: no statement about any block
  uint32_t bounce;               // cost ranking only (RT_TABLE_BOUNCE): bit 0 the sphere reflects, bit 1 it refracts (depth >= 3)
  uint32_t heavy;                // ... and its weight if it does BOTH (every hit a two-child node of the ray tree, main.js:268-278), else 0
};

// a sphere's screen rectangle on the workgroup grid and what a block that shows it is expected to cost
struct rt_cost_rect { double ys0, ys1; uint32_t tx0, tx1, weight, pad; };

struct rt_table_params {
  double as0, as1, as2;          // camera axis sums (main.js:187-191, quirk q1)
  double cam[3];
  double proj_w, proj_h, proj_d; // of the sample grid
  double epsilon;
  double lights[2][3];
  uint32_t tiles_x, ny, rb_per_tile;
  uint32_t tile_rows, tile_first, tile_stride, n_tiles;
  uint32_t ss, rows_per_wg, wg_w, wg_h;     // output rows / samples a workgroup covers
  uint32_t w, h;
  uint32_t n_balls, n_lights, n_rects, flags;
  uint32_t cost_bins;            // a block's cost lies in 1 .. cost_bins (1 when the launch is not ranked)
  uint32_t pad;
};

// first SAMPLE row of row block y of the launch
RT_HD inline double rt_block_row0(const rt_table_params &P, uint32_t y) {
  const uint32_t tile_i = y / P.rb_per_tile, rb = y - tile_i * P.rb_per_tile;
  return ((double)(P.tile_first + (uint64_t)tile_i * P.tile_stride) * P.tile_rows + (double)rb * P.rows_per_wg) * P.ss;
}

RT_HD inline uint32_t rt_block_cost(const rt_table_params &P, const rt_cost_rect *rects, uint32_t x, uint32_t y) {
  const double row0 = rt_block_row0(P, y);
  uint32_t cost = 2u;                                  // (sky runs are 1: the last class of a ranked table is theirs alone - a compact band's blocks are then 0 .. n - 1)
  for (uint32_t j = 0; j < P.n_rects; j++) {
    const rt_cost_rect &r = rects[j];
    if (x < r.tx0 || x > r.tx1) continue;
    if (row0 + P.wg_h <= r.ys0 || row0 > r.ys1) continue;
    cost += r.weight;
  }
  return cost < RT_COST_MAX ? cost : RT_COST_MAX;
}

// The statement of block (x, y) in pieces, so that the host (one block after the other, rt_block_statement below) and the device
// (rt_tables_gpu.hip: eight work-items per block, one sphere each) run the SAME operations on the same values and state the same words.

// the cone of the block's primary rays: unit axis, cos / sin of the half-angle; hit: the block counts as touched whatever the spheres
// say; doubt: nothing can be said about it (no geometry, or a box wider than a half space: never at these fields of view)
struct rt_cone { double ax[3], cos_a, sin_a; uint32_t hit, doubt; };

RT_HD inline rt_cone rt_block_cone(const rt_table_params &P, uint32_t x, uint32_t y) {
  rt_cone K;
  K.ax[0] = K.ax[1] = K.ax[2] = 0.0; K.cos_a = 1.0; K.sin_a = 0.0; K.hit = 1u; K.doubt = 1u;
  if (!(P.flags & RT_TABLE_GEOMETRY)) return K;
  const double row0 = rt_block_row0(P, y);
  const double Y1 = (P.proj_h - 0.5) - row0, Y0 = Y1 - (double)(P.wg_h - 1u);
  const double X0 = (double)((uint64_t)x * P.wg_w) + (0.5 - P.proj_w), X1 = X0 + (double)(P.wg_w - 1u);
  const double cx[4] = {X0, X1, X0, X1}, cy[4] = {Y0, Y0, Y1, Y1};
  double u[4][3], ax[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < 4; k++) {                      // the reference's ray: (s0 * X, s1 * Y, s2 * D), main.js:186-193 (q1)
    const double dx = P.as0 * cx[k], dy = P.as1 * cy[k], dz = P.as2 * P.proj_d, il = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
    u[k][0] = dx * il; u[k][1] = dy * il; u[k][2] = dz * il;
    for (int c = 0; c < 3; c++) ax[c] += u[k][c];
  }
  const double al = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  bool hit = !(al > 1e-3);                           // (a box wider than a half space)
  double cos_a = 1.0;
  if (!hit) {
    const double ial = 1.0 / al;
    for (int c = 0; c < 3; c++) ax[c] *= ial;
    for (int k = 0; k < 4; k++) cos_a = fmin(cos_a, ax[0] * u[k][0] + ax[1] * u[k][1] + ax[2] * u[k][2]);
    cos_a = fmax(cos_a - 1e-9, 0.0);                 // a slightly wider cone
    hit = !(cos_a > 1e-3);
  }
  K.ax[0] = ax[0]; K.ax[1] = ax[1]; K.ax[2] = ax[2];
  K.cos_a = cos_a; K.sin_a = sqrt(fmax(0.0, 1.0 - cos_a * cos_a));
  K.hit = K.doubt = hit ? 1u : 0u;
  return K;
}

// Cost ranking only: what a block's MIRRORS add to its cost.  The scene's dearest pixels are the hits on a sphere that both reflects and
// refracts (a binary ray tree: up to 31 nodes at depth 8) - and, after them, the pixels of OTHER spheres in which such a sphere is
// mirrored (measured, profiles/r04_ab_log.md: in the reference's own scene those blocks' waves ran 100-150 us, were ranked with the
// ordinary mirrors, started 100-140 us into a 200 us launch and WERE its last 70 us).  For candidate sphere ci of the block that
// reflects (or refracts): where the cone's axis meets it, the mirrored (or straight-through) direction there, a spread - the cone's
// half-angle magnified by the curvature, everything when the axis only grazes the sphere - and for every heavy sphere j the same
// angle test as everywhere else in this file.  A hit adds half of j's weight.  An estimate that only RANKS: no margin is owed,
// but host and device must compute the same number (plain binary64, no contraction, IEEE sqrt and division).
RT_HD inline uint32_t rt_bounce_cost(const rt_table_params &P, const rt_cone &K, const rt_ball *balls, uint32_t ci) {
  const rt_ball &B = balls[ci];
  if (!B.bounce || B.everywhere) return 0u;
  const double *ax = K.ax;
  const double cs = ax[0] * B.c[0] + ax[1] * B.c[1] + ax[2] * B.c[2], c = B.len * cs, disc = c * c - B.k;
  const bool graze = !(B.k > 0.0) || !(disc > 0.0);
  const double t = graze ? B.tangent : c - sqrt(disc);
  const double Q[3] = {P.cam[0] + t * ax[0], P.cam[1] + t * ax[1], P.cam[2] + t * ax[2]};
  double n[3] = {Q[0] - B.o[0], Q[1] - B.o[1], Q[2] - B.o[2]};
  const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (!(nl > 0.0)) return 0u;
  n[0] /= nl; n[1] /= nl; n[2] /= nl;
  const double cosi = -(ax[0] * n[0] + ax[1] * n[1] + ax[2] * n[2]);
  const bool wide = graze || !(cosi > 0.1);
  const double s_sig = wide ? 1.0 : fmin(1.0, K.sin_a * (1.0 + 2.0 * t / (B.R * cosi)) * 1.5 + 0.02), c_sig = sqrt(1.0 - s_sig * s_sig);
  const double s_thr = fmin(1.0, s_sig + 0.5), c_thr = sqrt(1.0 - s_thr * s_thr);          // straight through a refracting sphere: bent by up to ~30 degrees
  const double dr[3] = {ax[0] + 2.0 * cosi * n[0], ax[1] + 2.0 * cosi * n[1], ax[2] + 2.0 * cosi * n[2]};
  uint32_t extra = 0u;
  for (uint32_t j = 0; j < P.n_balls; j++) {
    const rt_ball &H = balls[j];
    if (!H.heavy || j == ci) continue;
    const double W[3] = {H.o[0] - Q[0], H.o[1] - Q[1], H.o[2] - Q[2]};
    const double wl = sqrt(W[0] * W[0] + W[1] * W[1] + W[2] * W[2]);
    if (!(wl > 0.0)) continue;
    const double sb = fmin(1.0, H.R / wl), cb = sqrt(1.0 - sb * sb);
    bool sees = false;
    if (B.bounce & 1u) sees = (dr[0] * W[0] + dr[1] * W[1] + dr[2] * W[2]) >= (c_sig * cb - s_sig * sb) * wl;
    if (!sees && (B.bounce & 2u)) sees = (ax[0] * W[0] + ax[1] * W[1] + ax[2] * W[2]) >= (c_thr * cb - s_thr * sb) * wl;
    if (sees) extra += H.heavy / 2u;
  }
  return extra;
}

// can some ray of the cone meet sphere B?  0: no; 1: yes (or NaN: touched); 2: the camera is inside / on / too near B - no statement
// about any block
RT_HD inline uint32_t rt_ball_touch(const rt_cone &K, const rt_ball &B) {
  if (B.everywhere) return 2u;
  const double cos_ab = K.cos_a * B.cos_b - K.sin_a * B.sin_b;          // cos(alpha + beta); alpha + beta < pi here
  const double cs = K.ax[0] * B.c[0] + K.ax[1] * B.c[1] + K.ax[2] * B.c[2];
  return (cs < cos_ab - 1e-7) ? 0u : 1u;             // untouched iff angle(axis, C) > alpha + beta, with a margin
}

// word 3 from the set of candidates (a bit per ball, at most 256; n_cand of them): count << 16 | loop index of the second << 8 |
// loop index of the first (in index order: the tie-break of the search), or 0
RT_HD inline uint32_t rt_cand_word(const rt_table_params &P, const rt_ball *balls, const uint64_t cand[4], uint32_t n_cand) {
  if (!(P.flags & RT_TABLE_CANDS) || n_cand == 0u || n_cand > 2u) return 0u;
  uint32_t idx[2] = {0u, 0u}, got = 0;
  for (uint32_t j = 0; j < P.n_balls && got < n_cand; j++) if (cand[j >> 6] >> (j & 63u) & 1ull) idx[got++] = balls[j].loop;
  uint32_t i0 = idx[0], i1 = n_cand > 1u ? idx[1] : 0u;
  if (n_cand > 1u && i1 < i0) { const uint32_t t = i0; i0 = i1; i1 = t; }
  return (n_cand << 16) | (i1 << 8) | i0;
}

// candidate sphere ci: which spheres can stand between the block's primary hits on it and light k - OR-ed into mk[k].  false: no
// statement can be made (the block's word is then 0xffffffff whatever the other candidates say)
RT_HD inline bool rt_cand_masks(const rt_table_params &P, const rt_cone &K, const rt_ball *balls, uint32_t ci, uint32_t mk[2]) {
  const rt_ball &B = balls[ci];
  const double *ax = K.ax;
  const double cos_a = K.cos_a, sin_a = K.sin_a;
  // the primary hits on sphere ci: distances [t1, t2] along rays within alpha of the axis
  const double cs = fmin(1.0, fmax(-1.0, ax[0] * B.c[0] + ax[1] * B.c[1] + ax[2] * B.c[2])), sn = sqrt(1.0 - cs * cs);
  const double c_hi = B.len * ((cs * cos_a + sn * sin_a >= 1.0 || sn <= sin_a) ? 1.0 : cs * cos_a + sn * sin_a);
  const double c_lo = fmax(B.len * (cs * cos_a - sn * sin_a), B.tangent);
  // (the kernel takes the FAR root when the near one lies within epsilon of the origin, main.js:431-436: a camera that close
  // to a sphere gets no statement)
  if (!(B.k > 0.0) || !(c_hi * c_hi >= B.k) || !(c_hi >= c_lo) || !(B.len - B.R >= 2.0 * fabs(P.epsilon))) return false;
  const double t1 = (c_hi - sqrt(fmax(c_hi * c_hi - B.k, 0.0))) * (1.0 - 1e-6), t2 = (c_lo - sqrt(fmax(c_lo * c_lo - B.k, 0.0))) * (1.0 + 1e-6);
  if (!(t2 >= t1) || !(t1 >= 0.0) || !(t2 <= 1.7976931348623157e308)) return false;
  const double m = 0.5 * (t1 + t2), rho = sqrt(0.25 * (t2 - t1) * (t2 - t1) + 2.0 * t2 * m * (1.0 - cos_a)) * (1.0 + 1e-6) + 1e-9 * t2;
  const double Q[3] = {P.cam[0] + m * ax[0], P.cam[1] + m * ax[1], P.cam[2] + m * ax[2]};
  const bool wide = (P.flags & RT_TABLE_WIDE) != 0u;
  for (uint32_t k = 0; k < P.n_lights; k++) {
    if (wide && mk[k] == 0xffffu) continue;        // many spheres: a light's set is "empty or not", and it already is not
    const double V[3] = {Q[0] - P.lights[k][0], Q[1] - P.lights[k][1], Q[2] - P.lights[k][2]};
    const double dist = sqrt(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]);
    // the patch seen from the light: a cone of half-angle asin(rho / dist) around V (the light inside the patch: every sphere)
    const bool patch_ok = (dist > rho * (1.0 + 1e-7)) && (dist <= 1.7976931348623157e308);
    const double s1 = patch_ok ? rho / dist : 0.0, c1 = sqrt(1.0 - s1 * s1), reach = (dist + rho) * (1.0 + 1e-7);
    for (uint32_t j = 0; j < P.n_balls; j++) {
      if (j == ci) continue;                       // a hit on sphere ci skips ci itself (main.js:294)
      const rt_ball &O = balls[j];
      bool inc = true;                             // light inside the occluder or the patch
      if (patch_ok && !O.always[k]) {
        // angle(V, W) <= asin(rho / dist) + asin(R / wl), as cosines scaled by |V| |W| (no division per sphere), and the occluder's
        // nearest point not beyond the patch
        const double cos12 = c1 * O.c2[k] - s1 * O.s2[k], vw = V[0] * O.lw[k][0] + V[1] * O.lw[k][1] + V[2] * O.lw[k][2];
        inc = !(vw < (cos12 - 1e-7) * (dist * O.wl[k])) && (O.wl_minus_R[k] <= reach);
      }
      if (inc) { if (wide) { mk[k] = 0xffffu; break; } mk[k] |= 1u << O.loop; }
    }
  }
  return true;
}

// touched / candidates / shadow masks of block (x, y), one block after the other (the host); *touched = 1 also when nothing can be said
RT_HD inline void rt_block_statement(const rt_table_params &P, const rt_ball *balls, uint32_t x, uint32_t y, uint32_t *touched, uint32_t *cands_out, uint32_t *smask_out,
                                     uint32_t *bounce_extra) {
  *touched = 1u; *cands_out = 0u; *smask_out = 0xffffffffu; *bounce_extra = 0u;
  const rt_cone K = rt_block_cone(P, x, y);
  bool hit = K.hit != 0u, doubt = K.doubt != 0u;
  if (!(P.flags & RT_TABLE_GEOMETRY)) return;
  // which spheres the cone can meet: a bit per ball (at most 256)
  uint64_t cand[4] = {0ull, 0ull, 0ull, 0ull};
  uint32_t n_cand = 0;
  for (uint32_t j = 0; j < P.n_balls && !doubt; j++) {
    const uint32_t t = rt_ball_touch(K, balls[j]);
    if (t == 2u) { hit = doubt = true; break; }
    if (t) { hit = true; cand[j >> 6] |= 1ull << (j & 63u); n_cand++; }
  }
  *touched = hit ? 1u : 0u;
  if (doubt || n_cand == 0u) return;
  *cands_out = rt_cand_word(P, balls, cand, n_cand);
  if (P.flags & RT_TABLE_BOUNCE)
    for (uint32_t ci = 0; ci < P.n_balls; ci++) if (cand[ci >> 6] >> (ci & 63u) & 1ull) *bounce_extra += rt_bounce_cost(P, K, balls, ci);
  if (!(P.flags & RT_TABLE_MASKS)) return;
  uint32_t mk[2] = {0u, 0u};
  for (uint32_t ci = 0; ci < P.n_balls; ci++) {
    if (!(cand[ci >> 6] >> (ci & 63u) & 1ull)) continue;
    if (!rt_cand_masks(P, K, balls, ci, mk)) return;
    if (mk[0] == 0xffffu && mk[1] == 0xffffu) return;       // many spheres, both sets not empty: the word is 0xffffffff whatever follows
  }
  *smask_out = mk[0] | (mk[1] << 16);
}

// the two words of an entry that describe WHERE its block is: word 0 = tile_x | rows_valid << 11 | first frame row << 15,
// word 1 = first row in this call's output band (| (run - 1) << 24 | sky << 31, added by the caller)
RT_HD inline void rt_block_place(const rt_table_params &P, uint32_t x, uint32_t y, uint32_t *w0, uint32_t *w1) {
  const uint32_t tile_i = y / P.rb_per_tile, rb = y - tile_i * P.rb_per_tile;
  const uint32_t trow0 = rb * P.rows_per_wg;                                             // first row of the block inside its tile
  const uint64_t frow0 = (uint64_t)(P.tile_first + (uint64_t)tile_i * P.tile_stride) * P.tile_rows + trow0;
  uint32_t rows_valid = 0;
  if (trow0 < P.tile_rows && frow0 < P.h) {
    rows_valid = P.rows_per_wg;
    if (P.tile_rows - trow0 < rows_valid) rows_valid = P.tile_rows - trow0;
    if (P.h - frow0 < rows_valid) rows_valid = (uint32_t)(P.h - frow0);
  }
  *w0 = (rows_valid << 11) | ((uint32_t)(frow0 < P.h ? frow0 : 0u) << 15) | x;              // frow0 < 65536 + 8: 17 bits
  *w1 = tile_i * P.tile_rows + trow0;                                                      // < 2^24 (checked by the caller): bits 24..31 are free
}

#endif
