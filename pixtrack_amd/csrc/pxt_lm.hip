// Fused Levenberg-Marquardt feature-metric pose refinement for gfx950 (MI355X).
//
// One PERSISTENT launch runs every iteration of every pyramid level: project the
// 3-D points, gather the 12-texel cross footprint of the 5 bilinear taps from the
// HWC query map, form residuals / robust + confidence weights / the 6x6 normal
// equations, solve the damped system, update the pose, test the stop criteria --
// with no host round trip (the reference path syncs twice per iteration: H,g to
// the CPU for Cholesky and the early-stop test).
//
// What it replaces: pixloc LearnedOptimizer._run / DirectAbsoluteCost /
// optimizer_step as driven by pixtrack/localization/pixloc_pose_refiners.py:260-262
// with pixtrack/optimizers/pixtrack_optimizer.py:5-18 (stop test every iteration)
// and pixtrack/localization/tracker.py:32-46 (masked-mean cost per iteration).
//
// Mapping to the machine
//  * A point is owned by a lane GROUP of LG lanes (8 for C<=32, 32 otherwise);
//    each lane owns 4 consecutive channels, so one texel of the HWC map is read
//    by the group as one contiguous LG*16 B segment (dwordx4 per lane).
//  * Because J = gradF (C x 2) * Jp (2 x 6), J^T J = Jp^T (gradF^T gradF) Jp:
//    only SIX scalars per point (sum r^2, gradF^T r (2), gradF^T gradF (3)) are
//    reduced across the group's lanes (xor butterflies); the 6+21 entries of g/H
//    are then formed once per point and accumulated in registers.
//  * Workgroups are persistent and co-resident (grid <= #CUs).  Per iteration
//    each publishes 32 floats (g, upper H, cost sum, valid count) as tagged 8-byte
//    granules with write-through (sc1) stores, and EVERY workgroup sweeps all
//    granules until their tags match (sc1 loads), reduces them in a fixed order and
//    redundantly solves the 6x6 system, so one store + one load round trip per
//    iteration is the only inter-workgroup synchronisation (cdna_hip_programming.md
//    G16 R2; results are bit-identical across workgroups and across runs).
#include "pxt_common.h"

#include <algorithm>

namespace pxt {

constexpr int kLmBlock = 512;
constexpr int kLmWaves = kLmBlock / PXT_WAVE;
constexpr int kLmMaxGrid = 256;
constexpr int kNAcc = 32;  // 6 g + 21 H + cost sum + n_valid + pad
constexpr int kGrpStride = 33;  // padded: leaders of one wave hit distinct banks
constexpr unsigned kSpinLimit = 1u << 22;  // polls of ~1.5 us before a spin gives up (pxt_lm_conf.spin_limit overrides)

struct LmLevelDev {
  const float* fmap;
  const float* fref;
  int h, w, C, cs;
  float cam[10];
  int ndist;
  float lambda[6];
};

// Which workgroups work on a problem: G of them, this one being number b.  The single-problem kernel's grid IS the
// problem's (G = gridDim.x, b = blockIdx.x); the batched kernel deals its grid to the problems (pxt_lm_refine_batch).
struct LmGrid {
  int G, b;
};

struct LmParams {
  const float* p3d;
  const uint8_t* mask;
  int n, n_levels;
  LmLevelDev lv[PXT_MAX_LEVELS];
  float T_init[12];
  pxt_lm_conf conf;
  float* out;
  float* log;
  unsigned long long* granules;  // [2][grid][kNAcc] {tag, value}
  unsigned* err;       // sticky error word
  // the next render's camera, derived from the final pose in the epilogue (pxt_lm_refine_cam)
  int cam_enabled;
  PoseConv cam_conv;
  float* cam_slot[2];
  float* cam_out;
};

__device__ inline void robust_loss(int kind, float alpha, float scale, float x, float& loss,
                                   float& w) {
  // pixloc losses.py: scaled_loss(x, fn, a) = (a^2 fn(x/a^2), fn'(x/a^2)).
  if (kind == 0) {
    loss = x;
    w = 1.f;
    return;
  }
  float a2 = scale * scale;
  float y = x / a2;
  float l, d;
  if (kind == 1) {  // huber
    if (y <= 1.f) {
      l = y;
      d = 1.f;
    } else {
      float sy = sqrtf(y);
      l = 2.f * sy - 1.f;
      d = fmaxf(1.1920929e-07f, 1.f / sy);
    }
  } else {  // barron(alpha)
    if (alpha == 0.f) {
      l = 2.f * log1pf(fminf(0.5f * y, 33e37f));
      d = 2.f / (y + 2.f);
    } else if (alpha == 2.f) {
      l = y;
      d = 1.f;
    } else {
      float beta = fmaxf(fabsf(alpha - 2.f), 1e-7f);
      float as = (alpha >= 0.f ? 1.f : -1.f) * fmaxf(fabsf(alpha), 1e-7f);
      l = 2.f * (beta / as) * (powf(y / beta + 1.f, 0.5f * alpha) - 1.f);
      d = powf(y / beta + 1.f, 0.5f * alpha - 1.f);
    }
  }
  loss = l * a2;
  w = d;
}

// Accumulates this workgroup's share of one LM iteration at one level.
// acc[0..5] = g, acc[6..26] = upper-triangular H (row-major), acc[27] = sum of
// valid robust costs, acc[28] = number of valid points.
// LG (lanes per point) is 8 or 32, wave-uniform at run time.
// Sum over the LG lanes of a point's group, every lane receiving the total.  The first four butterfly
// steps are DPP moves inside a 16-lane row (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror: for values
// that are already uniform over the smaller group a mirror is as good as an xor); only the 32-lane step
// crosses rows (one ds_bpermute).  (Six sums x five dependent __shfl_xor = 3.2k cycles per point round with
// hipcc's ds_bpermute lowering; stamps.)
__device__ inline float lm_dpp_add(float v, int ctrl_tag) {
  const int iv = __builtin_bit_cast(int, v);
  int o;
  if (ctrl_tag == 0) o = __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if (ctrl_tag == 1) o = __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if (ctrl_tag == 2) o = __builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false);  // row_half_mirror
  else o = __builtin_amdgcn_update_dpp(iv, iv, 0x140, 0xF, 0xF, false);                     // row_mirror
  return v + __builtin_bit_cast(float, o);
}

__device__ inline float lm_group_sum(float v, bool wide) {
  v = lm_dpp_add(v, 0);
  v = lm_dpp_add(v, 1);
  v = lm_dpp_add(v, 2);
  if (wide) {
    v = lm_dpp_add(v, 3);
    v += __shfl_xor(v, 16, PXT_WAVE);
  }
  return v;
}

// The same for a compile-time group width of 8, 16 or 32 lanes.
template <int LG>
__device__ inline float lm_group_sum_t(float v) {
  v = lm_dpp_add(v, 0);
  v = lm_dpp_add(v, 1);
  v = lm_dpp_add(v, 2);
  if (LG >= 16) v = lm_dpp_add(v, 3);
  if (LG >= 32) v += __shfl_xor(v, 16, PXT_WAVE);
  return v;
}

// Sum over the 16 lanes of a DPP row, every lane receiving the total (a fixed tree: the same bits in every workgroup).
__device__ inline float lm_row16_sum(float v) {
  v = lm_dpp_add(v, 0);
  v = lm_dpp_add(v, 1);
  v = lm_dpp_add(v, 2);
  return lm_dpp_add(v, 3);
}

// One point's contribution to g (acc[0..5]), the upper triangle of H (acc[6..26]), the cost sum and the valid count,
// from the six group-reduced scalars A = gradF^T r (2), B = gradF^T gradF (3) and its robust weight:
// J = gradF (C x 2) * Jp (2 x 6)  =>  J^T r = Jp^T A,  J^T J = Jp^T B Jp.
__device__ inline void lm_add_point(float* acc, float wgt, float rcost, const float* Jw, float px, float py, float pz,
                                    float A0, float A1, float B00, float B01, float B11) {
  // Jp = d(u,v)/d(delta) = Jw (2x3) * [I | -[p]x] (3x6), translation columns first.
  float J0[6], J1[6];
  J0[0] = Jw[0]; J0[1] = Jw[1]; J0[2] = Jw[2];
  J1[0] = Jw[3]; J1[1] = Jw[4]; J1[2] = Jw[5];
  // -[p]x = [[0, pz, -py], [-pz, 0, px], [py, -px, 0]]
  J0[3] = -Jw[1] * pz + Jw[2] * py;
  J0[4] = Jw[0] * pz - Jw[2] * px;
  J0[5] = -Jw[0] * py + Jw[1] * px;
  J1[3] = -Jw[4] * pz + Jw[5] * py;
  J1[4] = Jw[3] * pz - Jw[5] * px;
  J1[5] = -Jw[3] * py + Jw[4] * px;
  float M0[6], M1[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    M0[k] = B00 * J0[k] + B01 * J1[k];
    M1[k] = B01 * J0[k] + B11 * J1[k];
    acc[k] += wgt * (J0[k] * A0 + J1[k] * A1);
  }
  int idx = 6;
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int l = k; l < 6; ++l) acc[idx++] += wgt * (J0[k] * M0[l] + J1[k] * M1[l]);
  acc[27] += rcost;
  acc[28] += 1.f;
}

// The same for a group that serves ONE point per iteration: the 29 values go straight to the group's LDS record
// (no accumulator array: passed through the variant switch it would live in scratch).
__device__ inline void lm_point_record(float* dst, bool leader, float wgt, float rcost, const float* Jw, float px, float py,
                                       float pz, float A0, float A1, float B00, float B01, float B11) {
  float J0[6], J1[6];
  J0[0] = Jw[0]; J0[1] = Jw[1]; J0[2] = Jw[2];
  J1[0] = Jw[3]; J1[1] = Jw[4]; J1[2] = Jw[5];
  J0[3] = -Jw[1] * pz + Jw[2] * py;
  J0[4] = Jw[0] * pz - Jw[2] * px;
  J0[5] = -Jw[0] * py + Jw[1] * px;
  J1[3] = -Jw[4] * pz + Jw[5] * py;
  J1[4] = Jw[3] * pz - Jw[5] * px;
  J1[5] = -Jw[3] * py + Jw[4] * px;
  if (!leader) return;
  float M0[6], M1[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    M0[k] = B00 * J0[k] + B01 * J1[k];
    M1[k] = B01 * J0[k] + B11 * J1[k];
    dst[k] = 0.f + wgt * (J0[k] * A0 + J1[k] * A1);
  }
  int idx = 6;
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int l = k; l < 6; ++l) dst[idx++] = 0.f + wgt * (J0[k] * M0[l] + J1[k] * M1[l]);
  dst[27] = rcost;
  dst[28] = 1.f;
}

__device__ inline void lm_accumulate(const LmParams& P, const LmGrid grid, const LmLevelDev& L, const float* T,
                                     float* acc, const int LG, unsigned long long* dbg = nullptr) {
  const bool wide = LG == 32;
  int dbg_round = 0;
  const int GPW = PXT_WAVE / LG;  // groups per wave
  const int lane = threadIdx.x & (PXT_WAVE - 1);
  const int sub = lane & (LG - 1);
  const int grp = wide ? (lane >> 5) : (lane >> 3);
  const int wave = threadIdx.x / PXT_WAVE;
  const int wave_global = grid.b * kLmWaves + wave;
  const int stride = grid.G * kLmWaves * GPW;
  const Cam cam = make_cam(L.cam, L.ndist);
  const int W = L.w, H = L.h, C = L.C, cs = L.cs;
  const float pad = (float)P.conf.pad;

#pragma unroll 1
  for (int base = wave_global * GPW; base < P.n; base += stride) {
    const int n = base + grp;
    bool valid = n < P.n;
    float X = 0.f, Y = 0.f, Z = 1.f;
    if (valid) {
      X = P.p3d[3 * n];
      Y = P.p3d[3 * n + 1];
      Z = P.p3d[3 * n + 2];
      if (P.mask) valid = P.mask[n] != 0;
    }
#if PXT_EXP_STAMPS
    if (dbg && dbg_round == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[8] = __builtin_amdgcn_s_memtime(); }
#endif
    const float px = T[0] * X + T[1] * Y + T[2] * Z + T[9];
    const float py = T[3] * X + T[4] * Y + T[5] * Z + T[10];
    const float pz = T[6] * X + T[7] * Y + T[8] * Z + T[11];
    float u, v, Jw[6];
    valid = project_point(cam, px, py, pz, u, v, Jw) && valid;
    valid = valid && (u >= pad) && (v >= pad) && (u <= (float)(W - 1) - pad) &&
            (v <= (float)(H - 1) - pad);
    if (!valid) continue;  // group-uniform: contributes nothing (weight 0, not counted)

    const float fu = floorf(u), fv = floorf(v);
    const int ix0 = (int)fu, iy0 = (int)fv;
    const float ax = u - fu, ay = v - fv;
    const float w00 = (1.f - ax) * (1.f - ay), w10 = ax * (1.f - ay), w01 = (1.f - ax) * ay,
                w11 = ax * ay;
    // Column / row indices of the 4x4 neighbourhood, clamped for addressing; texels
    // outside the map count as zero (grid_sample padding_mode='zeros').
    int xi[4], yi[4];
    float xm[4], ym[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int xx = ix0 - 1 + k, yy = iy0 - 1 + k;
      xm[k] = (xx >= 0 && xx < W) ? 1.f : 0.f;
      ym[k] = (yy >= 0 && yy < H) ? 1.f : 0.f;
      xi[k] = min(max(xx, 0), W - 1);
      yi[k] = min(max(yy, 0), H - 1);
    }
    const float* row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) row[k] = L.fmap + (size_t)yi[k] * W * cs;

    float s_cost = 0.f, A0 = 0.f, A1 = 0.f, B00 = 0.f, B01 = 0.f, B11 = 0.f;
#pragma unroll 1
    for (int c0 = 4 * sub; c0 < C; c0 += 4 * LG) {
      // 12-texel cross footprint: rows 0,3 use columns 1,2; rows 1,2 use columns 0..3.
      float4 t01 = *(const float4*)(row[0] + (size_t)xi[1] * cs + c0);
      float4 t02 = *(const float4*)(row[0] + (size_t)xi[2] * cs + c0);
      float4 t10 = *(const float4*)(row[1] + (size_t)xi[0] * cs + c0);
      float4 t11 = *(const float4*)(row[1] + (size_t)xi[1] * cs + c0);
      float4 t12 = *(const float4*)(row[1] + (size_t)xi[2] * cs + c0);
      float4 t13 = *(const float4*)(row[1] + (size_t)xi[3] * cs + c0);
      float4 t20 = *(const float4*)(row[2] + (size_t)xi[0] * cs + c0);
      float4 t21 = *(const float4*)(row[2] + (size_t)xi[1] * cs + c0);
      float4 t22 = *(const float4*)(row[2] + (size_t)xi[2] * cs + c0);
      float4 t23 = *(const float4*)(row[2] + (size_t)xi[3] * cs + c0);
      float4 t31 = *(const float4*)(row[3] + (size_t)xi[1] * cs + c0);
      float4 t32 = *(const float4*)(row[3] + (size_t)xi[2] * cs + c0);
      float4 fr = *(const float4*)(L.fref + (size_t)n * cs + c0);
      const float m01 = ym[0] * xm[1], m02 = ym[0] * xm[2];
      const float m10 = ym[1] * xm[0], m11 = ym[1] * xm[1], m12 = ym[1] * xm[2],
                  m13 = ym[1] * xm[3];
      const float m20 = ym[2] * xm[0], m21 = ym[2] * xm[1], m22 = ym[2] * xm[2],
                  m23 = ym[2] * xm[3];
      const float m31 = ym[3] * xm[1], m32 = ym[3] * xm[2];
#define PXT_LM_CH(q)                                                                          \
  {                                                                                           \
    const float a01 = t01.q * m01, a02 = t02.q * m02, a10 = t10.q * m10, a11 = t11.q * m11,   \
                a12 = t12.q * m12, a13 = t13.q * m13, a20 = t20.q * m20, a21 = t21.q * m21,   \
                a22 = t22.q * m22, a23 = t23.q * m23, a31 = t31.q * m31, a32 = t32.q * m32;   \
    const float F = w00 * a11 + w10 * a12 + w01 * a21 + w11 * a22;                            \
    const float Fxp = w00 * a12 + w10 * a13 + w01 * a22 + w11 * a23;                          \
    const float Fxm = w00 * a10 + w10 * a11 + w01 * a20 + w11 * a21;                          \
    const float Fyp = w00 * a21 + w10 * a22 + w01 * a31 + w11 * a32;                          \
    const float Fym = w00 * a01 + w10 * a02 + w01 * a11 + w11 * a12;                          \
    const float gx = 0.5f * (Fxp - Fxm), gy = 0.5f * (Fyp - Fym);                             \
    const float r = F - fr.q;                                                                 \
    s_cost += r * r;                                                                          \
    A0 += r * gx;                                                                             \
    A1 += r * gy;                                                                             \
    B00 += gx * gx;                                                                           \
    B01 += gx * gy;                                                                           \
    B11 += gy * gy;                                                                           \
  }
      PXT_LM_CH(x) PXT_LM_CH(y) PXT_LM_CH(z) PXT_LM_CH(w)
#undef PXT_LM_CH
    }
    // Confidence: bilinear sample of channel C (same address for the whole group).
    const float q11 = row[1][(size_t)xi[1] * cs + C] * (ym[1] * xm[1]);
    const float q12 = row[1][(size_t)xi[2] * cs + C] * (ym[1] * xm[2]);
    const float q21 = row[2][(size_t)xi[1] * cs + C] * (ym[2] * xm[1]);
    const float q22 = row[2][(size_t)xi[2] * cs + C] * (ym[2] * xm[2]);
    const float wq = w00 * q11 + w10 * q12 + w01 * q21 + w11 * q22;
    const float wref = L.fref[(size_t)n * cs + C];

#if PXT_EXP_STAMPS
    if (dbg && dbg_round == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[9] = __builtin_amdgcn_s_memtime(); }
#endif
    s_cost = lm_group_sum(s_cost, wide);
    A0 = lm_group_sum(A0, wide);
    A1 = lm_group_sum(A1, wide);
    B00 = lm_group_sum(B00, wide);
    B01 = lm_group_sum(B01, wide);
    B11 = lm_group_sum(B11, wide);

    float rcost, wl;
    robust_loss(P.conf.loss, P.conf.loss_alpha, P.conf.loss_scale, s_cost, rcost, wl);
    const float wgt = wl * (wref * wq);

    lm_add_point(acc, wgt, rcost, Jw, px, py, pz, A0, A1, B00, B01, B11);
#if PXT_EXP_STAMPS
    if (dbg && dbg_round == 0) dbg[10] = __builtin_amdgcn_s_memtime();
    ++dbg_round;
#endif
  }
#if PXT_EXP_STAMPS
  if (dbg) { dbg[11] = __builtin_amdgcn_s_memtime(); dbg[12] = dbg_round; }
#endif
}

// ---- one-round levels: a group keeps ITS point for the whole level ------------------------------------------
// When the level's points fit the grid's groups (2341 points against 128 workgroups x 8 waves x 64 / LG lanes), every
// group serves at most one point, so what does not change between iterations - the point, its mask bit, its reference
// descriptor and confidence - is loaded ONCE per level and kept in registers: an iteration's memory chain is then
// pose (LDS) -> projection -> the 12-texel footprint, one round trip instead of two.  LG is picked per level so that
// one round suffices (C = 128: 32 lanes while n <= 2048 groups, then 16 lanes x 2 channel quads, then 8 x 4), and
// the points are dealt round-robin to the workgroups (point n -> workgroup n mod G), so every CU carries the same
// number of points whatever n is (the contiguous deal left 54 of 128 workgroups without a point at n = 2341, LG = 16).
template <int NF>
struct LmPoint {
  float X, Y, Z, wref;
  float4 fr[NF];  // the reference descriptor stays in registers while it is 1-2 quads per lane (else it is re-read)
  int n;
  bool valid;
};

// local group index of this lane's group within the workgroup, spread over the waves first
template <int LG>
__device__ inline int lm_local_group() {
  constexpr int GPW = PXT_WAVE / LG;
  const int lane = threadIdx.x & (PXT_WAVE - 1), wave = threadIdx.x / PXT_WAVE;
  return (lane / LG) * kLmWaves + wave;  // < GPW * kLmWaves
}

template <int LG, int CI>
__device__ inline void lm_load_point(const LmParams& P, const LmGrid grid, const LmLevelDev& L, LmPoint<2>& pt) {
  const int sub = (threadIdx.x & (PXT_WAVE - 1)) & (LG - 1);
  pt.n = lm_local_group<LG>() * grid.G + grid.b;
  pt.valid = pt.n < P.n;
  pt.X = pt.Y = 0.f;
  pt.Z = 1.f;
  pt.wref = 0.f;
  pt.fr[0] = pt.fr[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pt.valid) {
    pt.X = P.p3d[3 * pt.n];
    pt.Y = P.p3d[3 * pt.n + 1];
    pt.Z = P.p3d[3 * pt.n + 2];
    if (P.mask) pt.valid = P.mask[pt.n] != 0;
    const float* fr = L.fref + (size_t)pt.n * L.cs;
    if (CI <= 2) {
#pragma unroll
      for (int k = 0; k < (CI <= 2 ? CI : 0); ++k) pt.fr[k] = *(const float4*)(fr + 4 * sub + 4 * LG * k);
    }
    pt.wref = fr[L.C];
  }
}

template <int LG, int CI>
__device__ inline void lm_accumulate_cached(const LmParams& P, const LmLevelDev& L, const float* T, const LmPoint<2>& pt,
                                            float* grp_records, unsigned long long* dbg = nullptr) {
  const int sub = (threadIdx.x & (PXT_WAVE - 1)) & (LG - 1);
  float* const dst = grp_records + (threadIdx.x / LG) * kGrpStride;  // this group's record (the leader writes it)
  const Cam cam = make_cam(L.cam, L.ndist);
  const int W = L.w, H = L.h, C = L.C, cs = L.cs;
  const float pad = (float)P.conf.pad;
#if PXT_EXP_STAMPS
  if (dbg) dbg[8] = __builtin_amdgcn_s_memtime();
#endif
  const float px = T[0] * pt.X + T[1] * pt.Y + T[2] * pt.Z + T[9];
  const float py = T[3] * pt.X + T[4] * pt.Y + T[5] * pt.Z + T[10];
  const float pz = T[6] * pt.X + T[7] * pt.Y + T[8] * pt.Z + T[11];
  float u, v, Jw[6];
  bool valid = project_point(cam, px, py, pz, u, v, Jw) && pt.valid;
  valid = valid && (u >= pad) && (v >= pad) && (u <= (float)(W - 1) - pad) && (v <= (float)(H - 1) - pad);
  if (valid) {  // group-uniform
    const float fu = floorf(u), fv = floorf(v);
    const int ix0 = (int)fu, iy0 = (int)fv;
    const float ax = u - fu, ay = v - fv;
    const float w00 = (1.f - ax) * (1.f - ay), w10 = ax * (1.f - ay), w01 = (1.f - ax) * ay, w11 = ax * ay;
    int xi[4], yi[4];
    float xm[4], ym[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = ix0 - 1 + k, yy = iy0 - 1 + k;
      xm[k] = (xx >= 0 && xx < W) ? 1.f : 0.f;
      ym[k] = (yy >= 0 && yy < H) ? 1.f : 0.f;
      xi[k] = min(max(xx, 0), W - 1);
      yi[k] = min(max(yy, 0), H - 1);
    }
    const float* row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) row[k] = L.fmap + (size_t)yi[k] * W * cs + 4 * sub;
    float s_cost = 0.f, A0 = 0.f, A1 = 0.f, B00 = 0.f, B01 = 0.f, B11 = 0.f;
    const float m01 = ym[0] * xm[1], m02 = ym[0] * xm[2];
    const float m10 = ym[1] * xm[0], m11 = ym[1] * xm[1], m12 = ym[1] * xm[2], m13 = ym[1] * xm[3];
    const float m20 = ym[2] * xm[0], m21 = ym[2] * xm[1], m22 = ym[2] * xm[2], m23 = ym[2] * xm[3];
    const float m31 = ym[3] * xm[1], m32 = ym[3] * xm[2];
    // confidence: bilinear sample of channel C (same address for the whole group)
    const float q11 = (row[1] - 4 * sub)[(size_t)xi[1] * cs + C] * (ym[1] * xm[1]);
    const float q12 = (row[1] - 4 * sub)[(size_t)xi[2] * cs + C] * (ym[1] * xm[2]);
    const float q21 = (row[2] - 4 * sub)[(size_t)xi[1] * cs + C] * (ym[2] * xm[1]);
    const float q22 = (row[2] - 4 * sub)[(size_t)xi[2] * cs + C] * (ym[2] * xm[2]);
    // every load of a pair of channel quads is issued before the first use (up to 24 dwordx4 in flight per lane)
    constexpr int CB = CI == 2 ? 2 : 1;
#pragma unroll 1
    for (int kb = 0; kb < CI; kb += CB) {
    float4 t[CB][12], frl[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) {
      const int c0 = 4 * LG * (kb + k);
      if (CI > 2) frl[k] = *(const float4*)(L.fref + (size_t)pt.n * cs + 4 * sub + c0);
      t[k][0] = *(const float4*)(row[0] + (size_t)xi[1] * cs + c0);
      t[k][1] = *(const float4*)(row[0] + (size_t)xi[2] * cs + c0);
      t[k][2] = *(const float4*)(row[1] + (size_t)xi[0] * cs + c0);
      t[k][3] = *(const float4*)(row[1] + (size_t)xi[1] * cs + c0);
      t[k][4] = *(const float4*)(row[1] + (size_t)xi[2] * cs + c0);
      t[k][5] = *(const float4*)(row[1] + (size_t)xi[3] * cs + c0);
      t[k][6] = *(const float4*)(row[2] + (size_t)xi[0] * cs + c0);
      t[k][7] = *(const float4*)(row[2] + (size_t)xi[1] * cs + c0);
      t[k][8] = *(const float4*)(row[2] + (size_t)xi[2] * cs + c0);
      t[k][9] = *(const float4*)(row[2] + (size_t)xi[3] * cs + c0);
      t[k][10] = *(const float4*)(row[3] + (size_t)xi[1] * cs + c0);
      t[k][11] = *(const float4*)(row[3] + (size_t)xi[2] * cs + c0);
    }
#pragma unroll
    for (int k = 0; k < CB; ++k) {
      const float4 fr = CI <= 2 ? pt.fr[CI <= 2 ? k : 0] : frl[k];
#define PXT_LM_CH(q)                                                                                              \
  {                                                                                                               \
    const float a01 = t[k][0].q * m01, a02 = t[k][1].q * m02, a10 = t[k][2].q * m10, a11 = t[k][3].q * m11,        \
                a12 = t[k][4].q * m12, a13 = t[k][5].q * m13, a20 = t[k][6].q * m20, a21 = t[k][7].q * m21,        \
                a22 = t[k][8].q * m22, a23 = t[k][9].q * m23, a31 = t[k][10].q * m31, a32 = t[k][11].q * m32;      \
    const float F = w00 * a11 + w10 * a12 + w01 * a21 + w11 * a22;                                                \
    const float Fxp = w00 * a12 + w10 * a13 + w01 * a22 + w11 * a23;                                              \
    const float Fxm = w00 * a10 + w10 * a11 + w01 * a20 + w11 * a21;                                              \
    const float Fyp = w00 * a21 + w10 * a22 + w01 * a31 + w11 * a32;                                              \
    const float Fym = w00 * a01 + w10 * a02 + w01 * a11 + w11 * a12;                                              \
    const float gx = 0.5f * (Fxp - Fxm), gy = 0.5f * (Fyp - Fym);                                                 \
    const float r = F - fr.q;                                                                                     \
    s_cost += r * r;                                                                                              \
    A0 += r * gx;                                                                                                 \
    A1 += r * gy;                                                                                                 \
    B00 += gx * gx;                                                                                               \
    B01 += gx * gy;                                                                                               \
    B11 += gy * gy;                                                                                               \
  }
      PXT_LM_CH(x) PXT_LM_CH(y) PXT_LM_CH(z) PXT_LM_CH(w)
#undef PXT_LM_CH
    }
    }
    const float wq = w00 * q11 + w10 * q12 + w01 * q21 + w11 * q22;
#if PXT_EXP_STAMPS
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[9] = __builtin_amdgcn_s_memtime(); }
#endif
    s_cost = lm_group_sum_t<LG>(s_cost);
    A0 = lm_group_sum_t<LG>(A0);
    A1 = lm_group_sum_t<LG>(A1);
    B00 = lm_group_sum_t<LG>(B00);
    B01 = lm_group_sum_t<LG>(B01);
    B11 = lm_group_sum_t<LG>(B11);
    float rcost, wl;
    robust_loss(P.conf.loss, P.conf.loss_alpha, P.conf.loss_scale, s_cost, rcost, wl);
    lm_point_record(dst, sub == 0, wl * (pt.wref * wq), rcost, Jw, px, py, pz, A0, A1, B00, B01, B11);
  } else if (sub == 0) {
#pragma unroll
    for (int i = 0; i < 29; ++i) dst[i] = 0.f;
  }
#if PXT_EXP_STAMPS
  if (dbg) { dbg[10] = dbg[11] = __builtin_amdgcn_s_memtime(); dbg[12] = 1; }
#endif
}

__device__ inline unsigned ld_relaxed_u32(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Damped 6x6 solve (pixloc optimizer_step): H += diag(clamp(diag(H)*lambda, 1e-6));
// Cholesky, falling back to pivoted LU if a pivot is not positive.
__device__ inline bool solve6(const float* tot, const float* lambda, bool ok, float* delta) {
  float Hm[6][6], g[6];
  int idx = 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    g[k] = tot[k];
#pragma unroll
    for (int l = k; l < 6; ++l) {
      Hm[k][l] = tot[idx];
      Hm[l][k] = tot[idx];
      ++idx;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) Hm[k][k] += fmaxf(Hm[k][k] * lambda[k], 1e-6f);
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      g[k] = 0.f;
#pragma unroll
      for (int l = 0; l < 6; ++l) Hm[k][l] = (k == l) ? 1.f : 0.f;
    }
  }
  // Cholesky H = L L^T, fully unrolled so every entry stays in a register.  This runs on ONE lane while the whole grid
  // waits, and its six pivots are a dependent chain: per pivot ONE hardware reciprocal square root refined by one
  // Newton step (relative error ~1e-7, as good as sqrtf followed by a division, a third of their ~25 dependent
  // instructions); the column and both substitutions multiply by it.
  float Lm[6][6], inv[6];
  bool chol_ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = Hm[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= Lm[j][k] * Lm[j][k];
    chol_ok = chol_ok && (d > 0.f);
    const float r0 = __builtin_amdgcn_rsqf(d);
    const float rj = r0 * (1.5f - (0.5f * d) * (r0 * r0));
    Lm[j][j] = d * rj;
    inv[j] = rj;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float s = Hm[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k];
      Lm[i][j] = s * inv[j];
    }
  }
  if (chol_ok) {
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float s = g[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= Lm[i][k] * y[k];
      y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      float s = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) s -= Lm[k][i] * delta[k];
      delta[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) delta[i] = -delta[i];
    return true;
  }
  // LU with partial pivoting on [H | g].
  float A[6][7];
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) A[i][j] = Hm[i][j];
    A[i][6] = g[i];
  }
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    float best = fabsf(A[c][c]);
    for (int r = c + 1; r < 6; ++r)
      if (fabsf(A[r][c]) > best) {
        best = fabsf(A[r][c]);
        piv = r;
      }
    if (piv != c)
      for (int j = 0; j < 7; ++j) {
        float tmp = A[c][j];
        A[c][j] = A[piv][j];
        A[piv][j] = tmp;
      }
    for (int r = c + 1; r < 6; ++r) {
      float f = A[r][c] / A[c][c];
      for (int j = c; j < 7; ++j) A[r][j] -= f * A[c][j];
    }
  }
  for (int i = 5; i >= 0; --i) {
    float s = A[i][6];
    for (int k = i + 1; k < 6; ++k) s -= A[i][k] * delta[k];
    delta[i] = s / A[i][i];
  }
  for (int i = 0; i < 6; ++i) delta[i] = -delta[i];
  return false;
}

// T <- (so3exp(dw), dt) @ T ; returns (dR deg, dt) of the step (pixloc Pose.magnitude).
__device__ inline void apply_delta(const float* delta, float* T, float& dR_deg, float& dt_mag) {
  const float wx = delta[3], wy = delta[4], wz = delta[5];
  const float theta = sqrtf(wx * wx + wy * wy + wz * wz);
  float Rd[9];
  if (theta < 1e-7f) {
    Rd[0] = 1.f; Rd[1] = -wz; Rd[2] = wy;
    Rd[3] = wz;  Rd[4] = 1.f; Rd[5] = -wx;
    Rd[6] = -wy; Rd[7] = wx;  Rd[8] = 1.f;
  } else {
    const float it = 1.0f / theta;
    const float x = wx * it, y = wy * it, z = wz * it;
    float s, c1;
    if (theta < 0.25f) {  // an LM step: series to theta^9 / theta^10 (truncation < 1e-14 relative), no range reduction
      const float q = theta * theta;
      s = theta * (1.f + q * (-1.f / 6.f + q * (1.f / 120.f + q * (-1.f / 5040.f + q * (1.f / 362880.f)))));
      c1 = q * (0.5f + q * (-1.f / 24.f + q * (1.f / 720.f + q * (-1.f / 40320.f + q * (1.f / 3628800.f)))));
    } else {
      s = sinf(theta);
      c1 = 1.f - cosf(theta);
    }
    // W = [[0,-z,y],[z,0,-x],[-y,x,0]],  W^2 = w w^T - I (unit w)
    Rd[0] = 1.f + c1 * (x * x - 1.f);
    Rd[1] = -s * z + c1 * (x * y);
    Rd[2] = s * y + c1 * (x * z);
    Rd[3] = s * z + c1 * (x * y);
    Rd[4] = 1.f + c1 * (y * y - 1.f);
    Rd[5] = -s * x + c1 * (y * z);
    Rd[6] = -s * y + c1 * (x * z);
    Rd[7] = s * x + c1 * (y * z);
    Rd[8] = 1.f + c1 * (z * z - 1.f);
  }
  float Tn[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      Tn[3 * i + j] = Rd[3 * i] * T[j] + Rd[3 * i + 1] * T[3 + j] + Rd[3 * i + 2] * T[6 + j];
    Tn[9 + i] = Rd[3 * i] * T[9] + Rd[3 * i + 1] * T[10] + Rd[3 * i + 2] * T[11] + delta[i];
  }
  for (int i = 0; i < 12; ++i) T[i] = Tn[i];
  const float tr = Rd[0] + Rd[4] + Rd[8];
  const float cs = fminf(fmaxf((tr - 1.f) * 0.5f, -1.f), 1.f);
  dR_deg = fabsf(acosf(cs)) / 3.14159265358979323846f * 180.f;
  dt_mag = sqrtf(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
}

#if PXT_EXP_STAMPS  // timing experiment (scripts/lm_stamps.py): s_memtime of workgroup 0's first lane at 8 points per iteration
__device__ unsigned long long pxt_lm_stamps[256 * 16];
#define LM_STAMP(k) do { if (grid.b == 0 && blockIdx.x == 0 && threadIdx.x == 0 && run.total_iters < 256) pxt_lm_stamps[run.total_iters * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LM_STAMP(k)
#endif

constexpr int kRedStride = kNAcc + 1;  // padded: the 16 strided partial sums of a slot read distinct banks

struct LmShared {
  float grp[(kLmBlock / 8) * kGrpStride];   // the block's groups' sums (leaders park them)
  float red[kLmMaxGrid * kRedStride];       // every workgroup's 32 sums of this epoch
  float tot[kNAcc];
  float T[12];
  int flags[4];  // 0 stop level, 1 failed, 2 abort
};

struct LmRun {
  unsigned epoch, base, launch_id, spin_limit;
  int total_iters;
  bool failed, aborted;
};

// One iteration after the accumulation: the group leaders' sums are in sh.grp (n_groups records).  Fold them, publish,
// sweep every workgroup's granules, fold those, solve, update the pose, test the stop criteria.  Returns true when the
// level is over (stop criteria met, failure or abort).  Four workgroup barriers.
__device__ inline bool lm_step(const LmParams& P, const LmGrid grid, const LmLevelDev& L, int li, int it, int n_groups,
                               LmShared& sh, LmRun& run) {
  const int tid = threadIdx.x;
  const int G = grid.G;
  const int slot = tid >> 4, part = tid & 15;  // 32 slots x 16 parts: a slot's parts are one 16-lane DPP row
  __syncthreads();  // (A) the leaders' records are in sh.grp
  LM_STAMP(1);
  // Publish: the data IS the flag (cdna_hip_programming.md G16 recipe R2).  Each of the workgroup's 32 sums travels as
  // ONE aligned 8-byte {tag = base + epoch + 1, value} granule, written through (sc1) with no drain, no counter, no
  // fence; EVERY workgroup then sweeps all G x 32 granules of this epoch until every tag matches: an iteration's only
  // inter-workgroup traffic is one store and one (re-read) load round trip.  Tags count on from the workspace's
  // previous launch (`base`); two areas alternate by epoch parity (a workgroup publishes epoch e + 1 only after it
  // has read every granule of epoch e).
  unsigned long long* const area = P.granules + (size_t)(run.epoch & 1u) * G * kNAcc;
  const unsigned long long tag = (unsigned long long)(run.base + run.epoch + 1u) << 32;
  {  // the workgroup's own sums in a fixed order: 16 strided partials per slot, then a fixed 16-lane DPP tree - the
     // row's first lane stores the granule itself (round 3 went through LDS twice more and one wave published)
    float v = 0.f;
    if (slot < 29)
      for (int q = part; q < n_groups; q += 16) v += sh.grp[q * kGrpStride + slot];
    v = lm_row16_sum(v);
    if (part == 0)
      __hip_atomic_store(area + (size_t)grid.b * kNAcc + slot, tag | (unsigned long long)__float_as_uint(v),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  LM_STAMP(2);
  LM_STAMP(3);
  {  // the sweep: every thread re-reads its own few granules (G * 32 / 512: 8 at G = 128) until they carry this epoch
    const int n_gran = G * kNAcc;
    unsigned spins = 0;
    for (int i0 = 0; i0 < n_gran; i0 += 4 * kLmBlock) {
      for (;;) {
        bool ok = true;
        unsigned long long x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * kLmBlock + tid;
          x[k] = i < n_gran ? __hip_atomic_load(area + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
          ok = ok && ((x[k] >> 32) == (tag >> 32));
        }
        if (ok) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * kLmBlock + tid;
            if (i < n_gran) sh.red[(i >> 5) * kRedStride + (i & 31)] = __uint_as_float((unsigned)x[k]);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        if (++spins > run.spin_limit || (((spins & 255u) == 0u) && ld_relaxed_u32(P.err) == run.launch_id)) {
          __hip_atomic_store(P.err, run.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sh.flags[2] = 1;  // (benign race: every writer stores 1)
          i0 = n_gran;
          break;
        }
      }
    }
  }
  __syncthreads();  // (B)
  LM_STAMP(4);
  if (sh.flags[2]) {
    run.aborted = true;
    return true;
  }
  LM_STAMP(5);
  {  // fold in a FIXED order (bit-identical in every workgroup and every run): 16 strided partials per slot, the DPP tree
    float v = 0.f;
    for (int g = part; g < G; g += 16) v += sh.red[g * kRedStride + slot];
    v = lm_row16_sum(v);
    if (part == 0) sh.tot[slot] = v;
  }
  __syncthreads();  // (C)
  LM_STAMP(6);
  if (tid == 0) {
    const float n_valid = sh.tot[28];
    const bool fl = (sh.flags[1] != 0) || (n_valid < (float)P.conf.min_valid);
    float delta[6];
    const bool chol = solve6(sh.tot, L.lambda, !fl, delta);
    float Tn[12];
    for (int i = 0; i < 12; ++i) Tn[i] = sh.T[i];
    float dR, dt;
    apply_delta(delta, Tn, dR, dt);
    for (int i = 0; i < 12; ++i) sh.T[i] = Tn[i];
    float gn = 0.f;
    for (int i = 0; i < 6; ++i) gn += sh.tot[i] * sh.tot[i];
    gn = sqrtf(gn);
    const bool small_step = (dt < P.conf.dt_stop) && (dR < P.conf.dR_stop);
    const bool small_grad = gn < P.conf.grad_stop;
    sh.flags[0] = (small_step || small_grad) ? 1 : 0;
    sh.flags[1] = fl ? 1 : 0;
    if (grid.b == 0 && P.log) {
      float* lg = P.log + ((size_t)li * P.conf.num_iters + it) * PXT_LM_LOG_STRIDE;
      lg[0] = sh.tot[27] / n_valid;
      lg[1] = n_valid;
      lg[2] = dR;
      lg[3] = dt;
      lg[4] = gn;
      lg[5] = chol ? 0.f : 1.f;
      lg[6] = 0.f;
      lg[7] = 0.f;
      for (int i = 0; i < 12; ++i) lg[8 + i] = Tn[i];
    }
  }
  __syncthreads();  // (D)
  LM_STAMP(7);
  ++run.epoch;
  ++run.total_iters;
  run.failed = sh.flags[1] != 0;
  return sh.flags[0] != 0 || run.failed;
}

// Every lane of a group holds the same sums: the group leader parks them in LDS for lm_step's fold.
template <int LG>
__device__ inline void lm_park(const float* acc, LmShared& sh) {
  if ((threadIdx.x & (LG - 1)) == 0) {
    float* dst = sh.grp + (threadIdx.x / LG) * kGrpStride;
#pragma unroll
    for (int i = 0; i < 29; ++i) dst[i] = acc[i];
  }
}

// The accumulation of one iteration in the level's variant: 0 the general path (any channel count, several rounds of
// points per group), 1..4 one-round levels with the point in registers (LG, channel quads per lane) = (8, 1), (32, 1),
// (16, 2), (8, 4).  One loop body for all of them, so that lm_step - with its solve - is instantiated once.
__device__ inline void lm_accumulate_variant(int variant, const LmParams& P, const LmGrid grid, const LmLevelDev& L,
                                             const float* T, const LmPoint<2>& pt, LmShared& sh, unsigned long long* dbg) {
  switch (variant) {
    case 1: lm_accumulate_cached<8, 1>(P, L, T, pt, sh.grp, dbg); break;
    case 2: lm_accumulate_cached<32, 1>(P, L, T, pt, sh.grp, dbg); break;
    case 3: lm_accumulate_cached<16, 2>(P, L, T, pt, sh.grp, dbg); break;
    case 4: lm_accumulate_cached<8, 4>(P, L, T, pt, sh.grp, dbg); break;
    default: {
      float acc[kNAcc];
#pragma unroll
      for (int i = 0; i < kNAcc; ++i) acc[i] = 0.f;
      const int LGr = (L.C <= 32) ? 8 : 32;
      lm_accumulate(P, grid, L, T, acc, LGr, dbg);
      if (LGr == 8) lm_park<8>(acc, sh); else lm_park<32>(acc, sh);
    }
  }
}

// One problem's whole refinement on the workgroups `grid` names (every thread of every one of them runs this).
__device__ __forceinline__ void lm_refine_problem(const LmParams& P, const LmGrid grid, LmShared& sh) {
  const int tid = threadIdx.x;
  const int G = grid.G;

  if (tid < 12) sh.T[tid] = P.T_init[tid];
  if (tid == 0) {
    sh.flags[0] = 0;
    sh.flags[1] = 0;
    sh.flags[2] = 0;
  }
  __syncthreads();

  // Tags continue where the workspace's previous launch stopped (control word 1, advanced by workgroup 0 at the end of
  // every launch - which it reaches only after every workgroup has published, hence read this word, at least once): what
  // earlier launches left in the granule areas never carries a tag of this one, so nothing has to be zeroed between
  // launches (the memset node before every launch was 5 us + a dependent-launch gap on the frame's serial chain).
  LmRun run;
  run.epoch = 0;
  run.base = ld_relaxed_u32(P.err + 1);
  run.launch_id = run.base + 1u;  // what the error word holds when THIS launch timed out
  run.spin_limit = P.conf.spin_limit > 0 ? (unsigned)P.conf.spin_limit : kSpinLimit;
  run.total_iters = 0;
  run.failed = false;
  run.aborted = false;
  const int groups32 = G * kLmWaves * 2, groups16 = groups32 * 2, groups8 = groups32 * 4;
  const bool one_round_ok = P.conf.path != 2;  // (2: always the general path - A/B and tests)

  for (int li = 0; li < P.n_levels && !run.failed && !run.aborted; ++li) {
    const LmLevelDev& L = P.lv[li];
    int variant = 0;
    if (one_round_ok && L.C == 32 && P.n <= groups8) variant = 1;
    else if (one_round_ok && L.C == 128 && P.n <= groups32) variant = 2;
    else if (one_round_ok && L.C == 128 && P.n <= groups16) variant = 3;
    else if (one_round_ok && L.C == 128 && P.n <= groups8) variant = 4;
    LmPoint<2> pt;
    pt.X = pt.Y = pt.Z = pt.wref = 0.f;
    pt.fr[0] = pt.fr[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    pt.n = 0;
    pt.valid = false;
    if (variant == 1) lm_load_point<8, 1>(P, grid, L, pt);
    else if (variant == 2) lm_load_point<32, 1>(P, grid, L, pt);
    else if (variant == 3) lm_load_point<16, 2>(P, grid, L, pt);
    else if (variant == 4) lm_load_point<8, 4>(P, grid, L, pt);
    const int n_groups = kLmBlock / (variant == 2 ? 32 : variant == 3 ? 16 : variant != 0 ? 8 : (L.C <= 32 ? 8 : 32));
    int iters_done = 0;
    for (int it = 0; it < P.conf.num_iters; ++it) {
      LM_STAMP(0);
      float T[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = sh.T[i];
      unsigned long long* dbg = nullptr;
#if PXT_EXP_STAMPS
      if (grid.b == 0 && blockIdx.x == 0 && threadIdx.x == 0 && run.total_iters < 256) dbg = pxt_lm_stamps + run.total_iters * 16;
#endif
      lm_accumulate_variant(variant, P, grid, L, T, pt, sh, dbg);
      const bool stop = lm_step(P, grid, L, li, it, n_groups, sh, run);
      if (run.aborted) break;
      ++iters_done;
      if (stop) break;
    }
    if (grid.b == 0 && tid == 0) P.out[16 + li] = (float)iters_done;
  }

  if (grid.b == 0 && tid == 0) {
    // The next launch's tags start above every tag of this one.  After a time-out that needs a margin: workgroup 0
    // leaves at epoch e, but a workgroup that lagged may still complete epoch e (every other granule is tagged by
    // then) and publish epoch e + 1 with tag base + e + 2 before it sees the error word - which a next launch
    // starting at base + e + 1 would have accepted as its own epoch 0.  It can get at most one epoch
    // further than the slowest of the others (it needs THEIR granules of that epoch), so 8 is generous; the host
    // zeroes the workspace after a time-out as well (optimizer.py).
    P.err[1] = run.base + run.epoch + 1u + (run.aborted ? 8u : 0u);
    for (int i = 0; i < 12; ++i) P.out[i] = sh.T[i];
    P.out[12] = run.failed ? 1.f : 0.f;
    P.out[13] = run.aborted ? (float)PXT_E_TIMEOUT : 0.f;
    P.out[14] = (float)run.total_iters;
    // out[15] flips to 1 once the record and the log (all written by this thread) are visible
    // system-wide: a host that keeps `out` in pinned memory can poll this word instead of
    // waiting on an event (saves the ~25 us wake-up on the frame's critical path).
    __hip_atomic_store(&P.out[15], 1.f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ... and then the camera of the render queued behind this launch (one thread, float64: ~1 us after the record
    // has left; the renderer's kernels start after this kernel ends, in stream order)
    // A refinement that failed or timed out leaves the slots alone and marks the record -1: whatever render was queued
    // behind this launch then runs on the slot's OLD camera, and a caller of the C ABI sees in the camera record itself
    // that it must not be used (the Python tracker drops it on `success == False` as before).
    if (P.cam_enabled && (run.failed || run.aborted)) {
      if (P.cam_out) __hip_atomic_store(&P.cam_out[12], -1.f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (P.cam_enabled) {
      float cam[12];
      pose_to_camera_f64(sh.T, P.cam_conv, cam);
      for (int k = 0; k < 2; ++k)
        if (P.cam_slot[k])
          for (int i = 0; i < 12; ++i) P.cam_slot[k][i] = cam[i];
      if (P.cam_out) {
        for (int i = 0; i < 12; ++i) P.cam_out[i] = cam[i];
        __hip_atomic_store(&P.cam_out[12], 1.f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

__global__ __launch_bounds__(kLmBlock) void lm_refine_kernel(const LmParams P) {
  __shared__ LmShared sh;
  lm_refine_problem(P, LmGrid{(int)gridDim.x, (int)blockIdx.x}, sh);
}

// K independent problems in ONE persistent launch (pxt_lm_refine_batch): workgroup i works on problem i mod K as that
// problem's workgroup i / K.  Every problem has its own granule areas, error word, stop test and output record, so the
// problems never wait for each other - a problem that stops early simply retires its workgroups - and an iteration's
// exchange (~2.5 us), single-lane solve (~1.1 us) and barriers are paid once per iteration for K problems instead of K
// times in a row.  The hardware deals consecutive workgroups round-robin to the 8 XCDs, so with K = 8 a problem's
// workgroups share one XCD (its L2 then holds that problem's maps); nothing depends on it.  The problems' parameter
// records live in device memory (8 x ~1 KB exceeds the kernel-argument segment): uniform scalar loads.
__global__ __launch_bounds__(kLmBlock) void lm_refine_batch_kernel(const LmParams* __restrict__ params, const int K,
                                                                   const int contiguous) {
  __shared__ LmShared sh;
  const int G = (int)gridDim.x / K;
  const int p = contiguous ? (int)blockIdx.x / G : (int)blockIdx.x % K;
  const int b = contiguous ? (int)blockIdx.x % G : (int)blockIdx.x / K;
  lm_refine_problem(params[p], LmGrid{G, b}, sh);
}

// ---------------------------------------------------------------------------
// Sparse reference observations (pixloc_pose_refiners.py:327-368 + A.4).
// ---------------------------------------------------------------------------
struct SampleLevelDev {
  const float* fmap;
  float* out;
  int h, w, C, cs;
  float cam[10];
  int ndist;
  int x0, y0, fw, fh;  // the map is the window [x0, x0 + w) x [y0, y0 + h) of an fw x fh level (fw = w, fh = h, x0 = y0 = 0: the whole)
};

struct SampleParams {
  const float* p3d;
  float T[12];
  int n, n_levels, pad, normalize;
  SampleLevelDev lv[PXT_MAX_LEVELS];
  uint8_t* valid;
};

template <int LG>
__device__ inline bool sample_level(const SampleParams& P, const SampleLevelDev& L, int n,
                                    bool active, float px, float py, float pz, int sub) {
  const Cam cam = make_cam(L.cam, L.ndist);
  const int W = L.w, H = L.h, C = L.C, cs = L.cs;
  const int FW = L.fw, FH = L.fh;  // the full level (projection, visibility, padded in-image test)
  const float pad = (float)P.pad;
  float u = 0.f, v = 0.f;
  bool valid = project_point(cam, px, py, pz, u, v, nullptr) && active;
  valid = valid && (u >= pad) && (v >= pad) && (u <= (float)(FW - 1) - pad) &&
          (v <= (float)(FH - 1) - pad);
  if (!active) return false;
  float* o = L.out + (size_t)n * cs;
  if (!valid) {
    for (int c0 = 4 * sub; c0 < cs; c0 += 4 * LG) *(float4*)(o + c0) = make_float4(0, 0, 0, 0);
    return false;
  }
  const float fu = floorf(u), fv = floorf(v);
  const int ix0f = (int)fu, iy0f = (int)fv;
  const float ax = u - fu, ay = v - fv;
  const int x1f = min(ix0f + 1, FW - 1), y1f = min(iy0f + 1, FH - 1);
  const float mx = (ix0f + 1 < FW) ? 1.f : 0.f, my = (iy0f + 1 < FH) ? 1.f : 0.f;
  const float w00 = (1.f - ax) * (1.f - ay), w10 = ax * (1.f - ay) * mx, w01 = (1.f - ax) * ay * my,
              w11 = ax * ay * mx * my;
  // window coordinates (the caller's window holds every texel a valid point reads; the clamps only keep a caller's
  // mistake from reading outside the buffer)
  const int ix0 = min(max(ix0f - L.x0, 0), W - 1), iy0 = min(max(iy0f - L.y0, 0), H - 1);
  const int x1 = min(max(x1f - L.x0, 0), W - 1), y1 = min(max(y1f - L.y0, 0), H - 1);
  const float* p00 = L.fmap + ((size_t)iy0 * W + ix0) * cs;
  const float* p10 = L.fmap + ((size_t)iy0 * W + x1) * cs;
  const float* p01 = L.fmap + ((size_t)y1 * W + ix0) * cs;
  const float* p11 = L.fmap + ((size_t)y1 * W + x1) * cs;
  float ss = 0.f;
  // first pass: sum of squares of the interpolated descriptor
  for (int c0 = 4 * sub; c0 < C; c0 += 4 * LG) {
    float4 a = *(const float4*)(p00 + c0), b = *(const float4*)(p10 + c0),
           c = *(const float4*)(p01 + c0), d = *(const float4*)(p11 + c0);
    float4 f;
    f.x = w00 * a.x + w10 * b.x + w01 * c.x + w11 * d.x;
    f.y = w00 * a.y + w10 * b.y + w01 * c.y + w11 * d.y;
    f.z = w00 * a.z + w10 * b.z + w01 * c.z + w11 * d.z;
    f.w = w00 * a.w + w10 * b.w + w01 * c.w + w11 * d.w;
    ss += f.x * f.x + f.y * f.y + f.z * f.z + f.w * f.w;
    *(float4*)(o + c0) = f;
  }
  ss = group_allreduce_sum<LG>(ss);
  if (P.normalize) {
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int c0 = 4 * sub; c0 < C; c0 += 4 * LG) {
      float4 f = *(float4*)(o + c0);
      f.x *= inv; f.y *= inv; f.z *= inv; f.w *= inv;
      *(float4*)(o + c0) = f;
    }
  }
  if (sub == 0) {
    o[C] = w00 * p00[C] + w10 * p10[C] + w01 * p01[C] + w11 * p11[C];
    for (int c = C + 1; c < cs; ++c) o[c] = 0.f;
  }
  return true;
}

__global__ __launch_bounds__(256) void sample_sparse_kernel(const SampleParams P) {
  constexpr int LG = 32;
  constexpr int GPW = PXT_WAVE / LG;
  const int lane = threadIdx.x & (PXT_WAVE - 1);
  const int sub = lane % LG, grp = lane / LG;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) / PXT_WAVE;
  const int n = wave_global * GPW + grp;
  const bool active = n < P.n;
  float X = 0.f, Y = 0.f, Z = 1.f;
  if (active) {
    X = P.p3d[3 * n];
    Y = P.p3d[3 * n + 1];
    Z = P.p3d[3 * n + 2];
  }
  const float* T = P.T;
  const float px = T[0] * X + T[1] * Y + T[2] * Z + T[9];
  const float py = T[3] * X + T[4] * Y + T[5] * Z + T[10];
  const float pz = T[6] * X + T[7] * Y + T[8] * Z + T[11];
  bool all_valid = active;
  for (int l = 0; l < P.n_levels; ++l)
    all_valid = sample_level<LG>(P, P.lv[l], n, active, px, py, pz, sub) && all_valid;
  if (active && sub == 0) P.valid[n] = all_valid ? 1 : 0;
}

}  // namespace pxt

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace pxt;

extern "C" int64_t pxt_lm_workspace_bytes(void) {
  return (int64_t)(2 * kLmMaxGrid * kNAcc) * sizeof(unsigned long long) + 256;
}

extern "C" int pxt_lm_refine(const float* p3d, const uint8_t* point_mask, int32_t n_points,
                             const pxt_lm_level* levels, int32_t n_levels, const float* T_init,
                             const pxt_lm_conf* conf, float* out, float* log, void* workspace,
                             void* stream) {
  return pxt_lm_refine_cam(p3d, point_mask, n_points, levels, n_levels, T_init, conf, out, log, workspace, nullptr, stream);
}

namespace {

// Host records -> the kernel's parameter block (shared by the single-problem and the batched entry).
int lm_fill_params(LmParams& P, const float* p3d, const uint8_t* point_mask, int32_t n_points, const pxt_lm_level* levels,
                   int32_t n_levels, const float* T_init, const pxt_lm_conf* conf, float* out, float* log, void* workspace,
                   const pxt_lm_camera* cam) {
  if (!p3d || !levels || !T_init || !conf || !out || !workspace) return PXT_E_ARG;
  if (n_levels < 1 || n_levels > PXT_MAX_LEVELS || n_points < 1) return PXT_E_ARG;
  if (conf->num_iters < 1 || conf->pad < 0) return PXT_E_ARG;
  P.p3d = p3d;
  P.mask = point_mask;
  P.n = n_points;
  P.n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    const pxt_lm_level& s = levels[l];
    if (!s.fmap || !s.fref || s.C < 4 || (s.C % 4) != 0 || (s.cstride % 4) != 0 ||
        s.cstride < s.C + 1 || s.h < 2 || s.w < 2)
      return PXT_E_ARG;
    if (((uintptr_t)s.fmap % 16) != 0 || ((uintptr_t)s.fref % 16) != 0) return PXT_E_ARG;
    if (s.ndist != 0 && s.ndist != 2 && s.ndist != 4) return PXT_E_ARG;
    LmLevelDev& d = P.lv[l];
    d.fmap = s.fmap;
    d.fref = s.fref;
    d.h = s.h; d.w = s.w; d.C = s.C; d.cs = s.cstride;
    for (int i = 0; i < 10; ++i) d.cam[i] = s.cam[i];
    d.ndist = s.ndist;
    for (int i = 0; i < 6; ++i) d.lambda[i] = s.lambda[i];
  }
  for (int i = 0; i < 12; ++i) P.T_init[i] = T_init[i];
  P.conf = *conf;
  P.out = out;
  P.log = log;
  P.cam_enabled = 0;
  P.cam_slot[0] = P.cam_slot[1] = nullptr;
  P.cam_out = nullptr;
  if (cam) {
    if (!cam->cam_slot[0] && !cam->cam_slot[1] && !cam->cam_out13) return PXT_E_ARG;
    P.cam_enabled = 1;
    P.cam_conv = make_pose_conv(cam->conv27);
    P.cam_slot[0] = cam->cam_slot[0];
    P.cam_slot[1] = cam->cam_slot[1];
    P.cam_out = cam->cam_out13;
  }
  char* ws = (char*)workspace;
  P.err = (unsigned*)ws;  // 256-byte control block, then the granule areas
  P.granules = (unsigned long long*)(ws + 256);
  return PXT_OK;
}

// Every workgroup of a problem spins until every granule of the epoch is tagged, so all of them must be resident at
// once: never more workgroups than the device (or the partition / CU mask this process sees) can hold.  Several
// trackers of one process may run their LM kernels side by side (tests: three), so a launch takes at most half of
// the resident slots.  Cached per thread and device.
int lm_resident_cap(int* cap, bool whole_device = false) {
  static thread_local int resident_cap[16] = {0};
  int dev_id = 0;
  PXT_HIP_CHECK(hipGetDevice(&dev_id));
  *cap = 1 << 30;
  if (dev_id < 0 || dev_id >= 16) return PXT_OK;
  if (resident_cap[dev_id] == 0) {
    int per_cu = 0, cus = 0;
    PXT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lm_refine_batch_kernel, kLmBlock, 0));
    PXT_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id));
    resident_cap[dev_id] = per_cu * cus > 1 ? per_cu * cus : 1;
  }
  *cap = whole_device ? resident_cap[dev_id] : std::max(1, resident_cap[dev_id] / 2);
  return PXT_OK;
}

// Pinned staging records of the batched entry, a ring of four per thread and device: a slot is reused only after the
// copy that read it has completed (its event), which by then is several launches old.
struct LmStageSlot {
  LmParams* host = nullptr;
  hipEvent_t copied = nullptr;
};
constexpr int kLmStageSlots = 4;

}  // namespace

extern "C" int pxt_lm_refine_cam(const float* p3d, const uint8_t* point_mask, int32_t n_points,
                                 const pxt_lm_level* levels, int32_t n_levels, const float* T_init,
                                 const pxt_lm_conf* conf, float* out, float* log, void* workspace,
                                 const pxt_lm_camera* cam, void* stream) {
  LmParams P;
  const int rc = lm_fill_params(P, p3d, point_mask, n_points, levels, n_levels, T_init, conf, out, log, workspace, cam);
  if (rc != PXT_OK) return rc;
  int grid = conf->n_workgroups;
  // 128 workgroups: scripts/bench_lm.py, round 4, N = 2341: 8.7 us per iteration at 64, 8.5 at 128, 8.5 at 192 (where every
  // C = 128 level fits 32-lane groups in one round: the larger sweep costs what the cheaper variant saves), 8.8 at 256
  if (grid <= 0) grid = 128;
  if (grid > kLmMaxGrid) grid = kLmMaxGrid;
  int cap = 0;
  if (const int rcap = lm_resident_cap(&cap)) return rcap;
  if (grid > cap) grid = cap;
  // (no memset: the polled words - granule tags, error word - are compared with values only this launch writes)
  hipLaunchKernelGGL(lm_refine_kernel, dim3(grid), dim3(kLmBlock), 0, (hipStream_t)stream, P);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int64_t pxt_lm_batch_workspace_bytes(int32_t n_problems) {
  if (n_problems < 1 || n_problems > PXT_LM_MAX_BATCH) return PXT_E_ARG;
  return (int64_t)((size_t)n_problems * sizeof(LmParams) + 255) / 256 * 256;
}

extern "C" int pxt_lm_refine_batch(const pxt_lm_problem* problems, int32_t n_problems, const pxt_lm_conf* conf,
                                   void* batch_workspace, void* stream) {
  if (!problems || !conf || !batch_workspace) return PXT_E_ARG;
  if (n_problems < 1 || n_problems > PXT_LM_MAX_BATCH) return PXT_E_ARG;
  const int K = n_problems;
  for (int a = 0; a < K; ++a)  // every problem spins on ITS OWN granules: two problems on one workspace would read each other's
    for (int b = a + 1; b < K; ++b)
      if (problems[a].workspace == problems[b].workspace || problems[a].out == problems[b].out) return PXT_E_ARG;
  static thread_local LmStageSlot stage[16][kLmStageSlots];
  static thread_local int stage_next[16] = {0};
  int dev_id = 0;
  PXT_HIP_CHECK(hipGetDevice(&dev_id));
  if (dev_id < 0 || dev_id >= 16) return PXT_E_ARG;
  LmStageSlot& slot = stage[dev_id][stage_next[dev_id]];
  stage_next[dev_id] = (stage_next[dev_id] + 1) % kLmStageSlots;
  if (!slot.host) {
    PXT_HIP_CHECK(hipHostMalloc((void**)&slot.host, PXT_LM_MAX_BATCH * sizeof(LmParams), hipHostMallocDefault));
    PXT_HIP_CHECK(hipEventCreateWithFlags(&slot.copied, hipEventDisableTiming));
  } else {
    PXT_HIP_CHECK(hipEventSynchronize(slot.copied));
  }
  for (int k = 0; k < K; ++k) {
    const pxt_lm_problem& q = problems[k];
    const int rc = lm_fill_params(slot.host[k], q.p3d, q.point_mask, q.n_points, q.levels_host, q.n_levels, q.T_init_host, conf,
                                  q.out, q.log, q.workspace, q.cam_host);
    if (rc != PXT_OK) return rc;
  }
  // Workgroups per problem: the chip's 256 CUs dealt to the problems, one workgroup per CU (a 32-workgroup grid runs an
  // iteration within 5 % of the 128-workgroup one: profiles/r04_bench_lm.log), a multiple of 8, at most 128.
  int per = conf->n_workgroups;
  if (per <= 0) {
    per = std::max(8, std::min(128, 256 / K / 8 * 8));
    // ... but never fewer than one ROUND of 8-lane point groups needs (64 per workgroup): a level whose points do not
    // fit the grid's lane groups falls back to the several-rounds path - 17 instead of 8.3 us per iteration at
    // N = 2341 on 32 workgroups (profiles/r04_bench_lm.log)
    int n_max = 0;
    for (int k = 0; k < K; ++k) n_max = std::max(n_max, problems[k].n_points);
    per = std::max(per, std::min(128, ((n_max + 63) / 64 + 7) / 8 * 8));
  }
  if (per > kLmMaxGrid) per = kLmMaxGrid;
  // A batched launch stands for K trackers and may fill the device (256 VGPRs per lane: one 8-wave workgroup per CU, 256
  // resident workgroups; K = 8 problems then get 32 each).  Another LM launch running beside it on another queue could
  // leave both partly resident - each spinning on members that cannot start - which the bounded spins turn into
  // PXT_E_TIMEOUT: a lock-step tracker is the device's only LM client (one process per GPU).
  int cap = 0;
  if (const int rcap = lm_resident_cap(&cap, true)) return rcap;
  // The problems' workgroups are interleaved in dispatch order, so a grid that does not fit the device would leave EVERY
  // problem partly resident, each spinning on members that never start: the whole grid must be resident at once.
  if (per * K > cap) per = std::max(1, cap / K);
  hipStream_t s = (hipStream_t)stream;
  PXT_HIP_CHECK(hipMemcpyAsync(batch_workspace, slot.host, (size_t)K * sizeof(LmParams), hipMemcpyHostToDevice, s));
  PXT_HIP_CHECK(hipEventRecord(slot.copied, s));
  static const int contiguous = [] { const char* e = getenv("PXT_LM_BATCH_MAP"); return e ? atoi(e) : 0; }();
  hipLaunchKernelGGL(lm_refine_batch_kernel, dim3(per * K), dim3(kLmBlock), 0, s, (const LmParams*)batch_workspace, K, contiguous);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_sample_sparse(const float* p3d, int32_t n_points, const float* T,
                                 const pxt_sample_level* levels, int32_t n_levels, int32_t pad,
                                 int32_t normalize, uint8_t* valid, void* stream) {
  if (!p3d || !T || !levels || !valid) return PXT_E_ARG;
  if (n_levels < 1 || n_levels > PXT_MAX_LEVELS || n_points < 1 || pad < 0) return PXT_E_ARG;
  SampleParams P;
  P.p3d = p3d;
  for (int i = 0; i < 12; ++i) P.T[i] = T[i];
  P.n = n_points;
  P.n_levels = n_levels;
  P.pad = pad;
  P.normalize = normalize;
  P.valid = valid;
  for (int l = 0; l < n_levels; ++l) {
    const pxt_sample_level& s = levels[l];
    if (!s.fmap || !s.out || s.C < 4 || (s.C % 4) != 0 || (s.cstride % 4) != 0 ||
        s.cstride < s.C + 1 || s.h < 2 || s.w < 2)
      return PXT_E_ARG;
    if (((uintptr_t)s.fmap % 16) != 0 || ((uintptr_t)s.out % 16) != 0) return PXT_E_ARG;
    SampleLevelDev& d = P.lv[l];
    d.fmap = s.fmap;
    d.out = s.out;
    d.h = s.h; d.w = s.w; d.C = s.C; d.cs = s.cstride;
    for (int i = 0; i < 10; ++i) d.cam[i] = s.cam[i];
    d.ndist = s.ndist;
    if (s.full_w > 0) {  // a window of the level
      if (s.full_h < 1 || s.x0 < 0 || s.y0 < 0 || s.x0 + s.w > s.full_w || s.y0 + s.h > s.full_h) return PXT_E_ARG;
      d.x0 = s.x0; d.y0 = s.y0; d.fw = s.full_w; d.fh = s.full_h;
    } else {
      d.x0 = 0; d.y0 = 0; d.fw = s.w; d.fh = s.h;
    }
  }
  const int groups_per_block = 256 / 32;
  const int grid = (n_points + groups_per_block - 1) / groups_per_block;
  hipLaunchKernelGGL(sample_sparse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

#if PXT_EXP_STAMPS
extern "C" int pxt_debug_lm_stamps(void* host, int64_t bytes) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pxt::pxt_lm_stamps), (size_t)bytes) == hipSuccess ? 0 : -1;
}
#endif
