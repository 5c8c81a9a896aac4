// rt_tables.cpp — the tables the host builds for the kernels: primary-ray cull rectangles, per-light shadow grids, the bounce
// table, per-sphere tile weights and the launch table.  Pure host logic (no HIP call, no state): what rt_scene_upload uploads
// and what the no-GPU probes of include/rt_hip.h (rt_scene_cull_rects, rt_scene_bounce_candidates, rt_scene_launch_table)
// return to the CPU tests.

#include <math.h>
#include <stdint.h>

#include <cmath>
#include <vector>

#include <string.h>

#include "rt_tables.h"

namespace rt_tables {

// ------------------------------------------------------------------------------------ primary-ray cull rectangles (host logic)
// A primary ray has direction (s0*X, s1*Y, s2*D) with s = axisX+axisY+axisZ per component (the reference's
// component-indexed target formula, main.js:187-191), X = x - w/2 + 0.5, Y = h/2 - y - 0.5, D = projD.
// For a sphere at C = origin - camera with radius R, the rays with a given X/D = xi all lie in the plane
// through the camera spanned by (s0*xi, 0, s2) and the y axis; that plane meets the sphere iff the sphere's
// centre is within R of it:  (C0*s2 - C2*s0*xi)^2 <= R^2 (s2^2 + s0^2 xi^2),  a quadratic in xi whose root
// interval bounds every pixel column whose LINE meets the sphere (a superset of the columns whose ray hits
// it).  Same for rows with (C1, s1).  Unbounded or doubtful cases return the whole axis.
static void axis_bounds(double c_axis, double c_z, double s_axis, double s_z, double r2, double *lo, double *hi) {
  *lo = -INFINITY; *hi = INFINITY;
  const double A = s_axis * s_axis * (c_z * c_z - r2);
  const double B = -2.0 * c_axis * c_z * s_axis * s_z;
  const double Cq = s_z * s_z * (c_axis * c_axis - r2);
  const double disc = B * B - 4.0 * A * Cq;
  if (!(A > 1e-12 * s_axis * s_axis * (c_z * c_z + r2)) || !(disc >= 0.0)) return;   // image unbounded along this axis (or degenerate)
  const double sq = sqrt(disc);
  const double x1 = (-B - sq) / (2.0 * A), x2 = (-B + sq) / (2.0 * A);
  if (!(x1 <= x2) || !std::isfinite(x1) || !std::isfinite(x2)) return;
  *lo = x1 - 1e-7 * (1.0 + fabs(x1));                 // margins far above rounding, far below a pixel (1/D >= 1.5e-5)
  *hi = x2 + 1e-7 * (1.0 + fabs(x2));
}

rt_geom cull_rect(const rt_scene_header *hd, const rt_sphere &o) {
  rt_geom r = {-INFINITY, INFINITY, -INFINITY, INFINITY};          // {x_lo, x_hi, y_lo, y_hi}
  const double c[3] = {o.origin[0] - hd->cam_origin[0], o.origin[1] - hd->cam_origin[1], o.origin[2] - hd->cam_origin[2]};
  const double s[3] = {hd->cam_axis_x[0] + hd->cam_axis_y[0] + hd->cam_axis_z[0], hd->cam_axis_x[1] + hd->cam_axis_y[1] + hd->cam_axis_z[1],
                       hd->cam_axis_x[2] + hd->cam_axis_y[2] + hd->cam_axis_z[2]};
  const double k = (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) - o.r2;
  if (!(k > 1e-9 * o.r2) || !(o.r2 > 0.0)) return r;              // camera inside / on / near the sphere: no bound
  axis_bounds(c[0], c[2], s[0], s[2], o.r2, &r.ox, &r.oy);
  axis_bounds(c[1], c[2], s[1], s[2], o.r2, &r.oz, &r.r2);
  return r;
}

// ------------------------------------------------------------------------------------ enclosing sphere (host logic)
// The sphere that has every other sphere, every light and the camera strictly inside it, with a margin far
// above rounding (a skybox), or ~0u.  Such a sphere never shadows anything and is the closest hit only of rays that hit
// nothing else, which lets the product kernel take it out of the per-ray loops (exact, not approximate).
uint32_t enclosing_sphere(const rt_scene_header *hd, const rt_sphere *ob, const double lights[][3]) {
  for (uint32_t e = 0; e < hd->n_objects; e++) {
    const double re = sqrt(ob[e].r2), lim = re * (1.0 - 1e-6);
    auto dist_to = [&](const double q[3]) { const double x = q[0] - ob[e].origin[0], y = q[1] - ob[e].origin[1], z = q[2] - ob[e].origin[2]; return sqrt(x * x + y * y + z * z); };
    bool ok = re > 0.0 && dist_to(hd->cam_origin) < lim;
    for (uint32_t k = 0; k < hd->n_lights && ok; k++) ok = dist_to(lights[k]) < lim;
    for (uint32_t j = 0; j < hd->n_objects && ok; j++) if (j != e) ok = dist_to(ob[j].origin) + sqrt(ob[j].r2) < lim;
    if (ok && hd->n_objects > 1) return e;
  }
  return ~0u;
}

// ------------------------------------------------------------------------------------ shadow grids (host logic)
// Buffer layout (all 8-byte units): NL headers of 16 doubles {frame rows x'[3], y'[3], z'[3], gx0, gy0, 1/cell_w,
// 1/cell_h, pad[3]}, then per light (RT_SGRID*RT_SGRID + 1) cells of `words` uint64 each: bit j of a cell = sphere j
// (loop order) may block a shadow ray whose hit point projects into that cell; the last cell holds every sphere and
// serves hit points behind the light's frame plane.  Border cells stand for the half-lines beyond the grid.
std::vector<uint64_t> build_shadow_grid(const rt_sphere *objs, uint32_t n_loop, uint32_t n_lights, const double lights[][3]) {
  const uint32_t G = RT_SGRID, words = (n_loop + 63u) / 64u, cells = G * G + 1u;
  std::vector<uint64_t> buf((size_t)n_lights * 16u + (size_t)n_lights * cells * words, 0ull);
  double *hdr = (double *)buf.data();
  uint64_t *masks = buf.data() + (size_t)n_lights * 16u;
  for (uint32_t k = 0; k < n_lights; k++) {
    const double *Lp = lights[k];
    // frame: z' looks from the light at the centroid of the sphere centres
    double cz[3] = {0, 0, 0};
    for (uint32_t j = 0; j < n_loop; j++) for (int c = 0; c < 3; c++) cz[c] += (objs[j].origin[c] - Lp[c]) / n_loop;
    double len = sqrt(cz[0] * cz[0] + cz[1] * cz[1] + cz[2] * cz[2]);
    double z[3] = {0, -1, 0};
    if (len > 1e-9 && std::isfinite(len)) for (int c = 0; c < 3; c++) z[c] = cz[c] / len;
    const double up[3] = {fabs(z[1]) < 0.9 ? 0.0 : 1.0, fabs(z[1]) < 0.9 ? 1.0 : 0.0, 0.0};
    double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
    len = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (int c = 0; c < 3; c++) x[c] /= len;
    const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    // per-sphere rectangles in (x'/z', y'/z')
    std::vector<rt_geom> rect(n_loop);
    std::vector<char> skip(n_loop, 0);
    double gx0 = INFINITY, gx1 = -INFINITY, gy0 = INFINITY, gy1 = -INFINITY;
    for (uint32_t j = 0; j < n_loop; j++) {
      const double c[3] = {objs[j].origin[0] - Lp[0], objs[j].origin[1] - Lp[1], objs[j].origin[2] - Lp[2]};
      const double cx = x[0] * c[0] + x[1] * c[1] + x[2] * c[2], cy = y[0] * c[0] + y[1] * c[1] + y[2] * c[2], cq = z[0] * c[0] + z[1] * c[1] + z[2] * c[2];
      const double r2 = objs[j].r2, r = sqrt(r2);
      rt_geom q = {-INFINITY, INFINITY, -INFINITY, INFINITY};
      if (cq + r * (1.0 + 1e-9) + 1e-9 < 0.0) { skip[j] = 1; rect[j] = q; continue; }   // wholly behind the light: cannot lie between it and a point in front
      const double kk = (cx * cx + cy * cy + cq * cq) - r2;
      if (kk > 1e-9 * r2 && r2 > 0.0) {
        axis_bounds(cx, cq, 1.0, 1.0, r2, &q.ox, &q.oy);
        axis_bounds(cy, cq, 1.0, 1.0, r2, &q.oz, &q.r2);
      }
      rect[j] = q;
      if (std::isfinite(q.ox)) gx0 = fmin(gx0, q.ox);
      if (std::isfinite(q.oy)) gx1 = fmax(gx1, q.oy);
      if (std::isfinite(q.oz)) gy0 = fmin(gy0, q.oz);
      if (std::isfinite(q.r2)) gy1 = fmax(gy1, q.r2);
    }
    if (!(gx0 < gx1)) { gx0 = -1.0; gx1 = 1.0; }
    if (!(gy0 < gy1)) { gy0 = -1.0; gy1 = 1.0; }
    gx0 = fmax(gx0, -8.0); gx1 = fmin(gx1, 8.0); gy0 = fmax(gy0, -8.0); gy1 = fmin(gy1, 8.0);
    if (!(gx0 < gx1)) { gx0 = -8.0; gx1 = 8.0; }
    if (!(gy0 < gy1)) { gy0 = -8.0; gy1 = 8.0; }
    const double inv_cw = G / (gx1 - gx0), inv_ch = G / (gy1 - gy0);
    double *hk = hdr + 16u * k;
    for (int c = 0; c < 3; c++) { hk[c] = x[c]; hk[3 + c] = y[c]; hk[6 + c] = z[c]; }
    hk[9] = gx0; hk[10] = gy0; hk[11] = inv_cw; hk[12] = inv_ch;
    auto cell_of = [&](double v, double g0, double inv) -> uint32_t {     // the kernel's own mapping
      const double f = fmin(fmax((v - g0) * inv, 0.0), (double)(G - 1));
      return (uint32_t)f;
    };
    uint64_t *mk = masks + (size_t)k * cells * words;
    for (uint32_t j = 0; j < n_loop; j++) {
      mk[(size_t)(G * G) * words + (j >> 6)] |= 1ull << (j & 63u);                    // the "every sphere" cell
      if (skip[j]) continue;
      const uint32_t ix0 = cell_of(rect[j].ox, gx0, inv_cw), ix1 = cell_of(rect[j].oy, gx0, inv_cw);
      const uint32_t iy0 = cell_of(rect[j].oz, gy0, inv_ch), iy1 = cell_of(rect[j].r2, gy0, inv_ch);
      for (uint32_t iy = iy0; iy <= iy1; iy++) for (uint32_t ix = ix0; ix <= ix1; ix++) mk[((size_t)iy * G + ix) * words + (j >> 6)] |= 1ull << (j & 63u);
    }
  }
  return buf;
}

// Bounce table (product kernel, scenes with many spheres).  A reflected or refracted ray starts on the sphere it
// just hit, so its origin lies in that sphere's ball B(C_i, r_i); its direction d falls in one cell of a cube map with
// RT_BGRID x RT_BGRID cells per face.  Entry (i, cell) is the bit set of the loop spheres j that SOME ray from a point of
// B(C_i, r_i) with a direction of the cell can meet in front of its origin:
//     the line through p in B(C_i, r_i) with direction d passes within r_j of C_j only if the line through C_i with
//     direction d passes within R = r_i + r_j of C_j, i.e. only if the angle between d and D = C_j - C_i is at most
//     asin(R / |D|) (or the spheres are within R of each other: every direction), and the sphere lies ahead only for that
//     branch (not the one around -D); a cell is the cone around its centre direction with the half-angle of its farthest
//     corner.
// Everything is widened (1e-9 relative on R, 1e-6 rad on the cone) so that the kernel's own rounding - its hit point is
// on the sphere only up to an ulp, its cell index comes from a 2^-24 reciprocal - cannot put a ray outside the set its
// entry describes.  The table only prunes the candidates of the closest-hit search; the tests themselves are unchanged.
// Layout: [n_objects][RT_BCELLS][words] uint64, bit j = loop sphere j (device order).
std::vector<uint64_t> build_bounce_table(const rt_sphere *objs, uint32_t n_objects, uint32_t n_loop) {
  const uint32_t K = RT_BGRID, cells = RT_BCELLS, words = (n_loop + 63u) / 64u;
  std::vector<uint64_t> tab((size_t)n_objects * cells * words, 0ull);
  // cell cones: centre direction, cos/sin of the half-angle (the kernel's mapping: face = 2*major axis + (negative), u/v =
  // the other two axes in x,y,z order, a = u/|major|, b = v/|major| in [-1,1], cell = floor((a+1)K/2))
  std::vector<double> cdir(3u * cells), ccos(cells), csin(cells);
  for (uint32_t f = 0; f < 6; f++) {
    const int m = (int)(f >> 1), ua = (m == 0) ? 1 : 0, va = (m == 2) ? 1 : 2;
    const double sgn = (f & 1u) ? -1.0 : 1.0;
    for (uint32_t ib = 0; ib < K; ib++) for (uint32_t ia = 0; ia < K; ia++) {
      const uint32_t c = f * K * K + ib * K + ia;
      const double a0 = -1.0 + 2.0 * ia / K, a1 = -1.0 + 2.0 * (ia + 1) / K, b0 = -1.0 + 2.0 * ib / K, b1 = -1.0 + 2.0 * (ib + 1) / K;
      double ctr[3] = {0, 0, 0};
      ctr[m] = sgn; ctr[ua] = 0.5 * (a0 + a1); ctr[va] = 0.5 * (b0 + b1);
      const double cl = sqrt(ctr[0] * ctr[0] + ctr[1] * ctr[1] + ctr[2] * ctr[2]);
      for (int k = 0; k < 3; k++) ctr[k] /= cl;
      double worst = 1.0;                                   // smallest cosine between the centre and a corner
      for (int q = 0; q < 4; q++) {
        double v[3] = {0, 0, 0};
        v[m] = sgn; v[ua] = (q & 1) ? a1 : a0; v[va] = (q & 2) ? b1 : b0;
        const double vl = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        worst = fmin(worst, (v[0] * ctr[0] + v[1] * ctr[1] + v[2] * ctr[2]) / vl);
      }
      const double half = acos(fmax(-1.0, fmin(1.0, worst))) + 1e-6;
      for (int k = 0; k < 3; k++) cdir[3u * c + k] = ctr[k];
      ccos[c] = cos(half); csin[c] = sin(half);
    }
  }
  // per (i, j): one branch-free pass over the cells (structure-of-arrays, no division: both sides scaled by |D|)
  std::vector<double> cx(cells), cy(cells), cz(cells);
  for (uint32_t c = 0; c < cells; c++) { cx[c] = cdir[3u * c]; cy[c] = cdir[3u * c + 1]; cz[c] = cdir[3u * c + 2]; }
  std::vector<uint8_t> hit(cells);
  for (uint32_t i = 0; i < n_objects; i++) {
    const double ri = sqrt(objs[i].r2);
    uint64_t *row = tab.data() + (size_t)i * cells * words;
    // rays leave a sphere only if it reflects or refracts (albedo[3] > 0 or albedo[4] > 0, main.js:233,246): the rows of
    // the others are never read
    if (!(objs[i].albedo[3] > 0.0) && !(objs[i].albedo[4] > 0.0)) continue;
    for (uint32_t j = 0; j < n_loop; j++) {
      const double D0 = objs[j].origin[0] - objs[i].origin[0], D1 = objs[j].origin[1] - objs[i].origin[1], D2 = objs[j].origin[2] - objs[i].origin[2];
      const double Ld = sqrt(D0 * D0 + D1 * D1 + D2 * D2);
      const double R = (ri + sqrt(objs[j].r2)) * (1.0 + 1e-9);
      const uint64_t bit = 1ull << (j & 63u);
      const size_t wj = j >> 6;
      const bool everywhere = !(R < Ld * (1.0 - 1e-9)) || !std::isfinite(R) || !std::isfinite(Ld);   // overlapping / containing / degenerate: all cells
      if (everywhere) {
        for (uint32_t c = 0; c < cells; c++) row[(size_t)c * words + wj] |= bit;
        continue;
      }
      const double sa = R / Ld, ca = sqrt(fmax(0.0, 1.0 - sa * sa));
      // angle(centre, D) <= alpha + half  <=>  cos(angle) >= cos(alpha + half); alpha, half in (0, pi/2), so the sum is < pi
      for (uint32_t c = 0; c < cells; c++) {
        const double dotp = cx[c] * D0 + cy[c] * D1 + cz[c] * D2;                       // |D| cos(angle)
        const double cos_sum = ca * ccos[c] - sa * csin[c], sin_sum = sa * ccos[c] + ca * csin[c];
        hit[c] = (uint8_t)((sin_sum <= 0.0) | (dotp >= Ld * (cos_sum - 1e-12)));
      }
      for (uint32_t c = 0; c < cells; c++) if (hit[c]) row[(size_t)c * words + wj] |= bit;
    }
  }
  return tab;
}


// ------------------------------------------------------------------------------------ tile cost weights (host logic)
void scene_tile_weights(const rt_scene_header *hd, const rt_sphere *ob, std::vector<rt_geom> *cull, std::vector<uint32_t> *weight) {
  cull->resize(hd->n_objects);
  weight->resize(hd->n_objects);
  for (uint32_t i = 0; i < hd->n_objects; i++) {
    (*cull)[i] = cull_rect(hd, ob[i]);
    const bool lit = ob[i].albedo[1] > 0.0 || ob[i].albedo[2] > 0.0, refl = ob[i].albedo[3] > 0.0, refr = ob[i].albedo[4] > 0.0;
    const uint32_t depth = hd->segs > 1 ? (hd->segs - 1 < 4 ? hd->segs - 1 : 4) : 0;
    uint32_t wgt = (lit ? 2u : 0u) + ((refl || refr) ? 3u * depth : 0u);
    // a sphere that only refracts keeps its rays for several bounces (in, total internal reflections, out) where a mirror's leave at
    // once: measured on the reference's glass sphere, 60-75 us per wave against a mirror's 10-20 (profiles/r04_ab_log.md)
    if (refr && !refl && hd->segs > 1) wgt = (lit ? 2u : 0u) + 6u * (hd->segs - 1 < 7 ? hd->segs - 1 : 7);
    if (refl && refr && hd->segs > 1) wgt += 8u * (1u << (hd->segs - 1 < 5 ? hd->segs - 1 : 5));
    (*weight)[i] = wgt;
  }
}


// ------------------------------------------------------------------------------------ launch table (host side)
// The product kernel runs on a FLAT grid and reads, per workgroup, one 16-byte entry {tile_x | rows_valid << 11 | first frame row <<
// 15, first row in the output band | (run - 1) << 24 | sky << 31, shadow masks, primary candidates} (rt_kernel.hip: rt_pixel_of) -
// the tile / row-block arithmetic of the plain grid done once, plus what can be told about a block in advance (rt_block.h: sky
// blocks, shadow masks, candidates).  That also decides the ORDER in which the hardware hands the blocks out.  In grid order a frame
// ends on whatever lies at the bottom right - for the reference's scenes the floor and the sphere that both reflects and refracts,
// the dearest blocks of all - and the last of them run alone on an otherwise idle chip.  For launches of many workgroups the blocks
// are therefore ranked by a cost estimate (the weights of the spheres whose screen rectangle - the primary-ray cull's - touches
// the block) and listed dearest first, so the launch ends on sky; equal costs keep the grid's order (neighbours stay neighbours).
// Every block is still rendered exactly once by exactly one workgroup: the picture cannot change, only the tail does (measured:
// profiles/r02_ab_log.md).  Consecutive sky blocks of one row block are ONE entry (a run of up to RT_SKY_RUN_MAX blocks that does not
// cross a multiple of RT_SKY_RUN_MAX: its workgroup stores the constant into each).
//
// The LIBRARY builds the table on the GPU (rt_tables_gpu.hip, one work-item per block, the same rt_block.h); what is here is the
// part both share - the per-(scene, camera, frame size, tile set) parameters - and the host build: the oracle of the CPU tests
// and of the -m gpu test that compares the two, and the no-GPU probe rt_scene_launch_table.

// small launches have no tail worth ranking, and for very large ones (cfg5 on one GPU: 4.2 M workgroups, an 18 ms kernel) the
// tail is noise; the COUNT variant and the A/B switch keep the grid's order as well
// (ranked == 2: a compact band's launch - its blocks must come before the sky runs - is ranked whatever its size, up to the same limit)
bool table_is_ranked(int ranked, uint64_t n_blocks) { return ranked && (n_blocks >= 4096u || ranked == 2) && n_blocks <= (1u << 20); }

int make_table_params(const rt_scene_header *hd, const rt_sphere *ob, const std::vector<rt_geom> &cull, const std::vector<uint32_t> &weight,
                      uint32_t w, uint32_t h, uint32_t ss, const rt_tiles *tiles, uint32_t tiles_x, uint32_t rb_per_tile, double proj_w, double proj_h, double proj_d,
                      int ranked, bool mark_sky, uint32_t sky_sphere, bool shadow_masks, bool name_candidates, const double lights[][3],
                      rt_table_params *P, std::vector<rt_ball> *balls, std::vector<rt_cost_rect> *rects) {
  const uint32_t ny = tiles->n_tiles * rb_per_tile;
  const uint64_t n64 = (uint64_t)tiles_x * ny;
  if (tiles_x > 2048u || n64 >= (1ull << 31) || n64 == 0) return -1;   // (the caller reports it)
  memset(P, 0, sizeof *P);
  P->as0 = hd->cam_axis_x[0] + hd->cam_axis_y[0] + hd->cam_axis_z[0]; P->as1 = hd->cam_axis_x[1] + hd->cam_axis_y[1] + hd->cam_axis_z[1];
  P->as2 = hd->cam_axis_x[2] + hd->cam_axis_y[2] + hd->cam_axis_z[2];
  for (int c = 0; c < 3; c++) P->cam[c] = hd->cam_origin[c];
  P->proj_w = proj_w; P->proj_h = proj_h; P->proj_d = proj_d; P->epsilon = hd->epsilon;
  P->tiles_x = tiles_x; P->ny = ny; P->rb_per_tile = rb_per_tile;
  P->tile_rows = tiles->tile_rows; P->tile_first = tiles->tile_first; P->tile_stride = tiles->tile_stride; P->n_tiles = tiles->n_tiles;
  P->ss = ss; P->rows_per_wg = ss == 2u ? 2u : RT_TILE_H;                              // output rows a workgroup covers
  P->wg_w = RT_TILE_W * ss; P->wg_h = P->rows_per_wg * ss;                              // ... and samples
  P->w = w; P->h = h;
  const uint32_t n_loop = hd->n_objects - (sky_sphere != ~0u ? 1u : 0u);
  const bool want_masks = shadow_masks && n_loop <= 256u && hd->n_lights >= 1u && hd->n_lights <= 2u && lights != nullptr;
  const bool want_cands = name_candidates && n_loop <= 256u;                 // (no light needed, and any enclosing sphere: it is outside the loops)
  // (ranking) blocks whose mirrors show a sphere that both reflects and refracts are nearly as dear as that sphere's own: rt_block.h, rt_bounce_cost
  bool want_bounce = false;
  if (table_is_ranked(ranked, n64) && hd->segs >= 3u)
    for (uint32_t j = 0; j < hd->n_objects; j++) if (j != sky_sphere && ob[j].albedo[3] > 0.0 && ob[j].albedo[4] > 0.0 && weight[j] >= 16u) want_bounce = true;
  const bool geometry = std::isfinite(P->as0) && std::isfinite(P->as1) && std::isfinite(P->as2) && P->as0 != 0.0 && P->as1 != 0.0 && P->as2 != 0.0 &&
                        std::isfinite(proj_d) && proj_d > 0.0;
  P->flags = (mark_sky ? RT_TABLE_SKY : 0u) | (want_masks ? RT_TABLE_MASKS : 0u) | (want_cands ? RT_TABLE_CANDS : 0u) | (n_loop > 16u ? RT_TABLE_WIDE : 0u) |
             (table_is_ranked(ranked, n64) ? RT_TABLE_RANK : 0u) | ((geometry && (mark_sky || want_masks || want_cands || want_bounce)) ? RT_TABLE_GEOMETRY : 0u) |
             ((geometry && want_bounce) ? RT_TABLE_BOUNCE : 0u);
  P->cost_bins = 1u;
  P->n_lights = want_masks ? hd->n_lights : 0u;
  for (uint32_t k = 0; k < P->n_lights; k++) for (int c = 0; c < 3; c++) P->lights[k][c] = lights[k][c];
  balls->clear();
  if (P->flags & RT_TABLE_GEOMETRY)
    for (uint32_t j = 0; j < hd->n_objects; j++) {
      if (j == sky_sphere) continue;
      rt_ball B;
      memset(&B, 0, sizeof B);
      for (int c = 0; c < 3; c++) { B.o[c] = ob[j].origin[c]; B.c[c] = ob[j].origin[c] - hd->cam_origin[c]; }
      B.len = sqrt(B.c[0] * B.c[0] + B.c[1] * B.c[1] + B.c[2] * B.c[2]);
      B.R = sqrt(ob[j].r2) * (1.0 + 1e-7);
      B.k = B.len * B.len - ob[j].r2;                    // with the TRUE radius: the tangent length sqrt(k) bounds the hit distances from above
      B.loop = (sky_sphere != ~0u && j > sky_sphere) ? j - 1u : j;       // index in the product kernel's loop order (enclosing sphere last)
      B.everywhere = (!(ob[j].r2 > 0.0) || !std::isfinite(B.len) || !std::isfinite(B.R) || !(B.len > B.R * (1.0 + 1e-7))) ? 1u : 0u;      // camera inside / on / unknown
      if (!B.everywhere) { for (int c = 0; c < 3; c++) B.c[c] /= B.len; B.sin_b = B.R / B.len; B.cos_b = sqrt(1.0 - B.sin_b * B.sin_b); }
      B.tangent = sqrt(fmax(B.k, 0.0));
      if (P->flags & RT_TABLE_BOUNCE) {
        B.bounce = (ob[j].albedo[3] > 0.0 ? 1u : 0u) | (ob[j].albedo[4] > 0.0 ? 2u : 0u);
        B.heavy = (B.bounce == 3u && weight[j] >= 16u) ? weight[j] : 0u;
      }
      for (uint32_t k = 0; k < P->n_lights; k++) {      // the sphere as an occluder seen from light k
        for (int c = 0; c < 3; c++) B.lw[k][c] = B.o[c] - lights[k][c];
        B.wl[k] = sqrt(B.lw[k][0] * B.lw[k][0] + B.lw[k][1] * B.lw[k][1] + B.lw[k][2] * B.lw[k][2]);
        B.always[k] = (!(B.wl[k] > B.R * (1.0 + 1e-7)) || !std::isfinite(B.wl[k])) ? 1u : 0u;
        B.wl_minus_R[k] = B.wl[k] - B.R;
        B.s2[k] = B.always[k] ? 0.0 : B.R / B.wl[k];
        B.c2[k] = sqrt(1.0 - B.s2[k] * B.s2[k]);
      }
      balls->push_back(B);
    }
  P->n_balls = (uint32_t)balls->size();
  // per sphere: its cull rectangle, in sample coordinates, on the workgroup grid (the cost of a block: rt_block.h)
  rects->clear();
  if (P->flags & RT_TABLE_RANK) {
    const double W = (double)w * ss, H = (double)h * ss;
    for (size_t j = 0; j < cull.size(); j++) {
      if (!weight[j]) continue;
      const rt_geom &r = cull[j];                    // {x_lo, x_hi, y_lo, y_hi} in units of 1/D; X = sx - W/2 + 0.5, Y = H/2 - sy - 0.5
      const double sx0 = r.ox * proj_d + proj_w - 0.5, sx1 = r.oy * proj_d + proj_w - 0.5;
      const double sy0 = proj_h - 0.5 - r.r2 * proj_d, sy1 = proj_h - 0.5 - r.oz * proj_d;        // y grows downwards
      if (!(sx1 >= 0.0) || !(sx0 <= W) || !(sy1 >= 0.0) || !(sy0 <= H)) continue;            // off screen
      rt_cost_rect q;
      q.tx0 = (uint32_t)(fmax(sx0, 0.0) / P->wg_w); q.tx1 = (uint32_t)fmin(fmin(sx1, W - 1.0) / P->wg_w, (double)(tiles_x - 1u));
      q.ys0 = fmax(sy0, 0.0); q.ys1 = fmin(sy1, H - 1.0);
      q.weight = weight[j]; q.pad = 0u;
      rects->push_back(q);
    }
  }
  P->n_rects = (uint32_t)rects->size();
  // the costs a block of this launch can have: 1 .. cost_bins (what the ranking's per-row histograms are sized by); a launch whose
  // histograms would be unreasonably large (tens of thousands of row blocks AND many dear spheres) keeps the grid's order
  if (P->flags & RT_TABLE_RANK) {
    uint64_t bound = 2u;
    for (const rt_cost_rect &q : *rects) bound += q.weight;
    if (P->flags & RT_TABLE_BOUNCE) {                   // every mirror of a block may show every heavy sphere
      uint64_t heavy = 0u, mirrors = 0u;
      for (const rt_ball &B : *balls) { heavy += B.heavy / 2u; mirrors += B.bounce ? 1u : 0u; }
      bound += heavy * mirrors;
    }
    P->cost_bins = (uint32_t)(bound < RT_COST_MAX ? bound : RT_COST_MAX);
    if ((uint64_t)P->cost_bins * ny > (8u << 20)) { P->flags &= ~RT_TABLE_RANK; P->cost_bins = 1u; rects->clear(); P->n_rects = 0u; }
  }
  return 0;
}

// (`cull` = per-sphere screen rectangles, `weight` = per-sphere cost weights; returns RT_ENTRY_WORDS words per slot, 8 * ceil(blocks / 8) slots;
// empty on a launch that is too large for the table)
std::vector<uint32_t> build_launch_table(const rt_scene_header *hd, const rt_sphere *ob, const std::vector<rt_geom> &cull, const std::vector<uint32_t> &weight,
                                         uint32_t w, uint32_t h, uint32_t ss,
                                         const rt_tiles *tiles, uint32_t tiles_x, uint32_t rb_per_tile, double proj_w, double proj_h, double proj_d, int ranked,
                                         bool mark_sky, uint32_t sky_sphere, bool shadow_masks, bool name_candidates, const double lights[][3], uint32_t *n_entries, uint32_t sky_part) {
  if (n_entries) *n_entries = 0;
  rt_table_params P;
  std::vector<rt_ball> balls;
  std::vector<rt_cost_rect> rects;
  if (make_table_params(hd, ob, cull, weight, w, h, ss, tiles, tiles_x, rb_per_tile, proj_w, proj_h, proj_d, ranked, mark_sky, sky_sphere, shadow_masks, name_candidates, lights,
                        &P, &balls, &rects)) return {};
  const uint32_t ny = P.ny, n = tiles_x * ny;
  const bool rank = (P.flags & RT_TABLE_RANK) != 0u;
  std::vector<uint32_t> touched(n), cands(n), smask(n), cost(n, 1u);
  for (uint32_t y = 0; y < ny; y++)
    for (uint32_t x = 0; x < tiles_x; x++) {
      const size_t at = (size_t)y * tiles_x + x;
      uint32_t extra = 0u;
      rt_block_statement(P, balls.data(), x, y, &touched[at], &cands[at], &smask[at], &extra);
      if (rank) { const uint32_t c = rt_block_cost(P, rects.data(), x, y) + extra; cost[at] = c < RT_COST_MAX ? c : RT_COST_MAX; }
    }
  // The entries: one per workgroup.  A block that shows a sphere is an entry of its own; consecutive sky blocks of one row
  // block are ONE entry (a run of up to RT_SKY_RUN_MAX blocks: its workgroup stores the constant into each).
  struct item { uint32_t y, x, run, cost; };
  std::vector<item> items;
  items.reserve(n);
  uint32_t cmax = 1;
  const bool sky = (P.flags & RT_TABLE_SKY) && (P.flags & RT_TABLE_GEOMETRY);
  for (uint32_t y = 0; y < ny; y++)
    for (uint32_t x = 0; x < tiles_x;) {
      const size_t at = (size_t)y * tiles_x + x;
      if (!sky || touched[at]) { items.push_back({y, x, 0u, cost[at]}); if (cost[at] > cmax) cmax = cost[at]; x++; continue; }
      // (a run never crosses a multiple of RT_SKY_RUN_MAX blocks: whether a block starts a run is then a LOCAL question - its left
      // neighbour, its own column - which is what lets the device build decide it per block)
      uint32_t run = 1;
      while (x + run < tiles_x && (x + run) % RT_SKY_RUN_MAX != 0u && !touched[at + run]) run++;
      items.push_back({y, x, run, 1u});                  // (sky: base cost, last in the ranked order)
      x += run;
    }
  const uint32_t n_items = (uint32_t)items.size();
  if (n_entries) *n_entries = n_items;
  // counting sort, dearest first; equal costs keep the grid's order (neighbours stay neighbours)
  std::vector<uint32_t> start(cmax + 2u, 0u);
  if (rank) {
    for (const item &it : items) start[cmax - it.cost + 1u]++;
    for (uint32_t c = 0; c <= cmax; c++) start[c + 1u] += start[c];
  }
  // workgroup b's entry sits at (b % 8) * n8 + b / 8, n8 = ceil(BLOCKS / 8): one contiguous part per XCD (workgroups are dealt
  // round-robin over the 8 XCDs), at a stride the host knows before the number of entries is known (the device build's is only
  // known on the device); the slots behind an XCD's last entry stay zero
  const uint32_t n8 = (n + 7u) / 8u;
  std::vector<uint32_t> table((size_t)n8 * 8u * RT_ENTRY_WORDS, 0u);
  uint32_t next = 0;
  const bool words23 = (P.flags & RT_TABLE_GEOMETRY) != 0u;
  for (const item &it : items) {
    uint32_t w0, w1;
    rt_block_place(P, it.x, it.y, &w0, &w1);
    const uint32_t b = rank ? start[cmax - it.cost]++ : next++;                               // the workgroup that renders this entry
    if (it.run ? sky_part == 1u : sky_part == 2u) continue;                                    // RT_FLAG_NO_SKY / RT_FLAG_SKY_ONLY: the slot stays zero
    const size_t at = (size_t)(b & 7u) * n8 + (b >> 3);
    const size_t blk = (size_t)it.y * tiles_x + it.x;
    table[RT_ENTRY_WORDS * at] = w0;
    table[RT_ENTRY_WORDS * at + 1u] = w1 | (it.run ? (0x80000000u | ((it.run - 1u) << 24)) : 0u);
    table[RT_ENTRY_WORDS * at + 2u] = (words23 && (P.flags & RT_TABLE_MASKS)) ? smask[blk] : 0xffffffffu;
    table[RT_ENTRY_WORDS * at + 3u] = (words23 && (P.flags & RT_TABLE_CANDS)) ? cands[blk] : 0u;
  }
  return table;
}


}  // namespace rt_tables
