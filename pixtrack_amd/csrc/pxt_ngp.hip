// instant-ngp style NeRF inference renderer for gfx950.
//
// Replaces `testbed.render(w, h, spp, linear=True)` and the state pixtrack sets around
// it (pixtrack/visualization/run_vis_on_poses.py:38-56, pixtrack/utils/ingp_utils.py:22-44).
// The algorithm (ray setup, jitter, cone stepping, occupancy skipping, tcnn hash grid,
// SH, fp16 MLPs, compositing) is specified by oracle/ngp_oracle.py, which this kernel
// matches operation for operation in the ray march (this file is compiled with
// -ffp-contract=off so no multiply-add of the march is fused differently from numpy).
//
// Mapping to the machine
//  * Rays are enumerated pass-fastest (ray id = pixel * spp + pass): the 8 passes of a pixel are neighbours in the
//    live list, and the per-pass results land in a [pixel][spp] buffer that `resolve` averages in a fixed order (no
//    atomics on the image, order-deterministic).
//  * A march thread finds ITS ray's next K occupied samples (bitfield DDA, divergent); a shade wave then evaluates
//    8 rays x 8 samples together:
//      - hash grid: 16 levels x 8 corners of 4-byte (2 x fp16) gathers per lane;
//      - both MLPs on v_mfma_f32_32x32x16_f16 with samples as the N (column) axis.
//        Hidden activations never leave registers: the D fragment of one layer,
//        ReLU'd and packed to fp16, IS the B fragment of the next layer once the
//        next layer's weight columns are permuted to the D row order (done once when
//        the weights are packed into fragment order).  Only the network inputs need
//        a cross-lane move: one v_permlane32_swap per dword places "lane = sample"
//        data into the 2 x 32-column operand layout.
//      - the 24 weight fragments (24 KiB) live in LDS, one conflict-free
//        ds_read_b128 per fragment per step.
#include "pxt_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace pxt {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// The file is compiled -ffp-contract=off: the MARCH is written operation for operation like oracle/ngp_oracle.py, so that
// kernel and oracle visit identical samples.  The hash-grid interpolation of a sample's features (ngp_encode_level*) may
// contract: 8 v_pk_fma_f32 instead of 8 v_pk_mul_f32 + 8 v_pk_add_f32 per level, one rounding less per accumulation; the
// two features are rounded to fp16 right after, so nearly every packed feature keeps its bits (every parity test and
// fixture unchanged: sample counts equal, masks bit-exact).  Shade launch -3 %, value +0.8 % (profiles/r05_experiments.md #19).
constexpr int kGrid = 128;
constexpr int kMaxLevels = 16;
constexpr int kNumFrags = 24;  // d1:4 d2:4 c1:4 c2:8 c3:4
constexpr int kFragD1 = 0, kFragD2 = 4, kFragC1 = 8, kFragC2 = 12, kFragC3 = 20;

struct NgpLevel {
  float scale;
  unsigned res, offset, size, hashed;
};

struct NgpParams {
  const unsigned* grid;        // [entries] packed 2 x fp16
  unsigned grid_bytes;
  const half8* wfrag;          // [kNumFrags][64] fragment-ordered weights
  const uint8_t* occ;          // bitfield
  NgpLevel lv[kMaxLevels];
  int n_levels, cascades;
  float aabb_scale, cone_angle, depth_scale, dt_lo, dt_hi;
  // view
  float cam[12];
  const float* cam_dev;  // when set, the camera is read from here (written on the device: pxt_ngp_render_both_from_pose)
  float focal, k1;
  float lo[3], hi[3];
  float bg[4];
  float min_T;
  int W, H, spp, mode;
  int coop;            // wave-cooperative corner fetch of the coarse levels (ngp_render_body); PXT_NGP_COOP=0 turns it off (A/B)
  int srgb_to_linear;  // Shade: a finished ray's colour goes through srgb_to_linear before the spp mean (model.linear_colors == 0)
  float* out;        // float RGBA of the render's own mode (mode 2: the Shade image); may be null when an 8-bit output stands in
  float* out_depth;  // mode 2: the Depth image (optional)
  uint8_t* out_u8;   // optional [H][W][3]: (rgb * 255).astype(uint8) of the Shade image (modes 0, 2)
  uint8_t* out_nz;   // optional [H][W]: uint8(depth * 255) != 0 of the Depth image (modes 1, 2): get_mask's plane
  unsigned long long* stats;
  long long enum_lo, enum_hi;  // the part of the ray enumeration this pipeline (slice) generates
};

__device__ inline float calc_dt(float t, float cone, float lo, float hi) {
  return fminf(fmaxf(t * cone, lo), hi);
}

// frexp exponent of a non-negative float (0 for 0; denormals flush to exponent -126: they only
// occur within 1e-38 of the cube centre, where every cascade index clamps to 0 anyway).
__device__ inline int frexp_exp(float v) {
  const unsigned b = __builtin_bit_cast(unsigned, v);
  const int e = (int)((b >> 23) & 0xffu);
  return e == 0 ? 0 : e - 126;
}

// 2^k as a float, k in [-126, 127] (exact; replaces ldexpf(1, k))
__device__ inline float pow2i(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }

__device__ inline int mip_from_pos(float x, float y, float z, int cascades) {
  float m = fmaxf(fabsf(x - 0.5f), fmaxf(fabsf(y - 0.5f), fabsf(z - 0.5f)));
  return min(max(frexp_exp(m) + 1, 0), cascades - 1);
}

__device__ inline unsigned pack_h2(float a, float b) {
  half2_t h;
  h[0] = (half_t)a;
  h[1] = (half_t)b;
  return __builtin_bit_cast(unsigned, h);
}

__device__ inline half8 as_half8(unsigned a, unsigned b, unsigned c, unsigned d) {
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(half8, v);
}

// vdst' = {lo lanes: vdst, hi lanes: src(lo lanes)};  src' = {lo lanes: vdst(hi lanes), hi: src}
__device__ inline void swap32(unsigned& vdst, unsigned& src) {
  auto r = __builtin_amdgcn_permlane32_swap(vdst, src, false, false);
  vdst = r[0];
  src = r[1];
}

__device__ inline void sh4_eval(float x, float y, float z, float* o) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// ReLU + fp16 pack of 8 consecutive accumulator rows -> one B fragment.  The ReLU runs on the
// packed halves (fp16 rounding is monotone and sign preserving, so max(fp16(x), 0) = fp16(max(x, 0))
// up to the sign of a zero, which an MFMA operand does not see): 4 v_pk_max_f16 instead of 8 v_max_f32.
__device__ inline half8 relu_pack8(const f32x16& a, int base, bool relu) {
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (half_t)a[base + j];
  if (relu) {
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    r = __builtin_elementwise_max(r, z);
  }
  return r;
}

// One hash-grid level at a warped position in [0,1]^3 -> packed (f0, f1) fp16 pair.
__device__ inline unsigned ngp_encode_level(const unsigned* __restrict__ grid, const NgpLevel& Lv, float ux,
                                            float uy, float uz) {
#pragma clang fp contract(fast)  // (this function only: see the note at kGrid)
  const float qx = ux * Lv.scale + 0.5f, qy = uy * Lv.scale + 0.5f, qz = uz * Lv.scale + 0.5f;
  const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
  const float ax = qx - fx, ay = qy - fy, az = qz - fz;
  const unsigned gx = (unsigned)(int)fx, gy = (unsigned)(int)fy, gz = (unsigned)(int)fz;
  unsigned vals[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const unsigned cx = gx + (c & 1), cy = gy + ((c >> 1) & 1), cz = gz + ((c >> 2) & 1);
    // tcnn: index % level_size.  A hashed level has exactly 2^log2_hashmap entries (mask), a
    // dense level's index is already < res^3 <= size: no integer division either way.
    unsigned idx;
    if (Lv.hashed)
      idx = ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (Lv.size - 1u);
    else
      idx = min(cx + cy * Lv.res + cz * Lv.res * Lv.res, Lv.size - 1u);
    vals[c] = grid[idx + Lv.offset];
  }
  float f0 = 0.f, f1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = 1.0f;
    w = w * ((c & 1) ? ax : (1.0f - ax));
    w = w * ((c & 2) ? ay : (1.0f - ay));
    w = w * ((c & 4) ? az : (1.0f - az));
    const half2_t hv = __builtin_bit_cast(half2_t, vals[c]);
    f0 += w * (float)hv[0];
    f1 += w * (float)hv[1];
  }
  return pack_h2(f0, f1);
}

typedef __attribute__((ext_vector_type(2))) unsigned uint2_t;
typedef __attribute__((ext_vector_type(2))) float float2_t;
// The trilinear sum of 8 packed (f0, f1) fp16 corners - corner c = x + 2 y + 4 z, weight ((1 * wx) * wy) * wz, f += w * v in
// corner order: the arithmetic of ngp_encode_level, with the eight weights formed as FOUR packed products of (x-pair) x
// (y) x (z) - v_pk_mul_f32 - instead of twelve scalar ones.  Every product has the same operands, every sum the same order:
// the same bits.
__device__ __forceinline__ unsigned ngp_trilinear_pk(const unsigned (&vals)[8], float ax, float ay, float az) {
#pragma clang fp contract(fast)  // (as ngp_encode_level: the f += w * v chain may contract)
  const float2_t wx = {1.0f - ax, ax};
  const float wy0 = 1.0f - ay, wz0 = 1.0f - az;
  const float2_t wxy[2] = {wx * wy0, wx * ay};
  float2_t f = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; c += 2) {
    const float2_t w = wxy[(c >> 1) & 1] * ((c & 4) ? az : wz0);  // corners c (x = 0) and c + 1 (x = 1)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const half2_t hv = __builtin_bit_cast(half2_t, vals[c + x]);
      const float2_t v = {(float)hv[0], (float)hv[1]};
      f += w[x] * v;
    }
  }
  return pack_h2(f[0], f[1]);
}
// The same level, for a caller whose level is wave-uniform (the shade kernel's rolled level loop): identical indices and
// arithmetic, with the integer work pared down - the two y and two z hash products (or row / plane
// offsets) are formed once and shared by the 8 corners (v_mul_lo_u32 is quarter rate), and the gathers
// are buffer loads with a 32-bit byte offset and the level's base as the scalar offset instead of 64-bit
// address arithmetic per corner (18 v_mad_u64_u32 + 14 v_lshl_add_u64 per item in the first version).
__device__ inline unsigned ngp_encode_level_uniform(const __amdgpu_buffer_rsrc_t grid, const NgpLevel& Lv, float ux,
                                                    float uy, float uz) {
#pragma clang fp contract(fast)  // (this function only: see the note at kGrid)
  const float qx = ux * Lv.scale + 0.5f, qy = uy * Lv.scale + 0.5f, qz = uz * Lv.scale + 0.5f;
  const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
  const float ax = qx - fx, ay = qy - fy, az = qz - fz;
  const unsigned gx = (unsigned)(int)fx, gy = (unsigned)(int)fy, gz = (unsigned)(int)fz;
  unsigned vals[8];
  const int base = (int)(Lv.offset * 4u);
  // byte offsets formed directly: ((a ^ b ^ c) & m) << 2 == ((a << 2) ^ (b << 2) ^ (c << 2)) & (m << 2), and a product's
  // shift is a product by 4 x the constant (mod 2^32) - eight shifts per level less
  const unsigned gx4[2] = {gx << 2, (gx << 2) + 4u};
  if (Lv.hashed) {
    const unsigned mask4 = (Lv.size - 1u) << 2;
    const unsigned hy[2] = {gy * (2654435761u * 4u), gy * (2654435761u * 4u) + 2654435761u * 4u};
    const unsigned hz[2] = {gz * (805459861u * 4u), gz * (805459861u * 4u) + 805459861u * 4u};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned off = (gx4[c & 1] ^ hy[(c >> 1) & 1] ^ hz[(c >> 2) & 1]) & mask4;
      vals[c] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(grid, (int)off, base, 0);
    }
  } else {
    const unsigned r4 = Lv.res * 4u, r24 = Lv.res * Lv.res * 4u, last4 = (Lv.size - 1u) << 2;
    const unsigned ry[2] = {gy * r4, gy * r4 + r4};
    const unsigned rz[2] = {gz * r24, gz * r24 + r24};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned off = min(gx4[c & 1] + ry[(c >> 1) & 1] + rz[(c >> 2) & 1], last4);
      vals[c] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(grid, (int)off, base, 0);
    }
  }
  return ngp_trilinear_pk(vals, ax, ay, az);
}

// Both MLPs for the wave's 64 samples (lane = sample).  Flo/Fhi: this lane's 32 encoded
// features as 16 packed dwords (levels 0..7 / 8..15).  All 64 lanes must call (MFMA).
// One 32-sample column block through both MLPs.  x0/x1: the block's two input B fragments,
// shb: its SH fragment.  Returns the raw density logit row and the three activated colour rows
// as held by this lane (rows 0..3 live in the low lane half).
template <bool DEPTH_ONLY>
__device__ inline void ngp_mlp_block(const half8* s_w, int lane, half8 x0, half8 x1, half8 shb, float& logit,
                                     float* rgb) {
  // ---- density MLP: 32 -> 64 (ReLU) -> 16 ----
  f32x16 h1[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    f32x16 a = {0};
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragD1 + 2 * rb) * 64 + lane], x0, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragD1 + 2 * rb + 1) * 64 + lane], x1, a, 0, 0, 0);
    h1[rb] = a;
  }
  f32x16 dout = {0};
#pragma unroll
  for (int q = 0; q < 4; ++q)
    dout = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragD2 + q) * 64 + lane],
                                                  relu_pack8(h1[q >> 1], 8 * (q & 1), true), dout, 0, 0, 0);
  logit = dout[0];
  if (DEPTH_ONLY) {
    rgb[0] = rgb[1] = rgb[2] = 0.f;
    return;
  }
  // ---- colour MLP: [16 density outputs | 16 SH] -> 64 -> 64 -> 16 ----
  f32x16 c1[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    f32x16 a = {0};
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC1 + 2 * rb) * 64 + lane], relu_pack8(dout, 0, false), a,
                                               0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC1 + 2 * rb + 1) * 64 + lane], shb, a, 0, 0, 0);
    c1[rb] = a;
  }
  half8 c1p[4];  // packed once, used by both output row blocks
#pragma unroll
  for (int q = 0; q < 4; ++q) c1p[q] = relu_pack8(c1[q >> 1], 8 * (q & 1), true);
  f32x16 c2[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    f32x16 a = {0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC2 + 4 * rb + q) * 64 + lane], c1p[q], a, 0, 0, 0);
    c2[rb] = a;
  }
  f32x16 cout = {0};
#pragma unroll
  for (int q = 0; q < 4; ++q)
    cout = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC3 + q) * 64 + lane],
                                                  relu_pack8(c2[q >> 1], 8 * (q & 1), true), cout, 0, 0, 0);
  // The activation is applied BEFORE any cross-lane move: v_permlane32_swap reading a register
  // an MFMA is still writing returned stale rows (regs > 0) on gfx950 / ROCm 7.2; a VALU op in
  // between gets the MFMA->VALU wait states the compiler does model.
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb[c] = 1.0f / (1.0f + expf(-cout[c]));
}

// Both MLPs for the wave's 64 samples (lane = sample).  Flo/Fhi: this lane's 32 encoded
// features as 16 packed dwords (levels 0..7 / 8..15).  All 64 lanes must call (MFMA).
// The two 32-sample column blocks run one after the other (sched_barrier keeps the compiler
// from interleaving them), which halves the live accumulators -> 4 waves per SIMD.
template <bool DEPTH_ONLY>
__device__ inline void ngp_mlp(const half8* s_w, int lane, unsigned* Flo, unsigned* Fhi, const unsigned* shB0,
                               const unsigned* shB1, float& logit, float* rgbv) {
#pragma unroll
  for (int i = 0; i < 8; ++i) swap32(Flo[i], Fhi[i]);
  float l0, l1, c0[3], c1[3];
  ngp_mlp_block<DEPTH_ONLY>(s_w, lane, as_half8(Flo[0], Flo[1], Flo[2], Flo[3]), as_half8(Flo[4], Flo[5], Flo[6], Flo[7]),
                            as_half8(shB0[0], shB0[1], shB0[2], shB0[3]), l0, c0);
  __builtin_amdgcn_sched_barrier(0);
  ngp_mlp_block<DEPTH_ONLY>(s_w, lane, as_half8(Fhi[0], Fhi[1], Fhi[2], Fhi[3]), as_half8(Fhi[4], Fhi[5], Fhi[6], Fhi[7]),
                            as_half8(shB1[0], shB1[1], shB1[2], shB1[3]), l1, c1);
  // rows 0..3 of a block sit in the low lane half: bring block 1's to the high lanes
  // (the logits pass through a VALU op first: same MFMA -> permlane hazard as above)
  unsigned r0 = __builtin_bit_cast(unsigned, fmaxf(l0, -1e30f)), r1 = __builtin_bit_cast(unsigned, fmaxf(l1, -1e30f));
  swap32(r0, r1);
  logit = __builtin_bit_cast(float, r0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    unsigned a0 = __builtin_bit_cast(unsigned, c0[c]), a1 = __builtin_bit_cast(unsigned, c1[c]);
    swap32(a0, a1);
    rgbv[c] = __builtin_bit_cast(float, a0);
  }
}

// Fused encode + MLPs (lane = sample) used by the straggler kernel and the point query.
template <bool DEPTH_ONLY>
__device__ inline void ngp_eval(const NgpParams& P, const half8* s_w, int lane, bool alive, float ux,
                                float uy, float uz, const unsigned* shB0, const unsigned* shB1,
                                float& logit, float* rgbv) {
  unsigned Flo[8], Fhi[8];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    unsigned pk = 0u;
    if (alive && l < P.n_levels) pk = ngp_encode_level(P.grid, P.lv[l], ux, uy, uz);
    if (l < 8) Flo[l] = pk; else Fhi[l - 8] = pk;
  }
  ngp_mlp<DEPTH_ONLY>(s_w, lane, Flo, Fhi, shB0, shB1, logit, rgbv);
}

// ===========================================================================
// The renderer: three launches per render (or per chain of K renders), all on the caller's stream.
//
//   raygen   : ngp_raygen_kernel - a ray per (pixel, spp pass), box test, start jitter, order-preserving compaction of the
//              rays that hit the render box into a list (ray id, start t, unit direction)
//   render   : ngp_render_kernel - PERSISTENT waves: a wave holds 8 rays of the list and repeats two steps until its
//              share of the list is used up: every ray's next K = 8 occupied lattice samples (ngp_march_group: the 8 lanes
//              of a ray probe 8 lattice points per trip), then the wave's 8 x 8 samples through the hash grid (16 levels x
//              8 corners of 4-byte gathers per lane), both MLPs on MFMA, in-order compositing; a ray that terminated or
//              left the box is replaced by the next one of the wave's share.  Samples travel from the march to the shade
//              step in registers, a ray's transmittance and colour stay in registers from its first sample to its last.
//   resolve  : fixed-order mean over spp + background (+ the 8-bit planes the tracking loop consumes)
//
// History (profiles/HISTORY.md, profiles/r06_experiments.md).  Rounds 1-5 ran a render as a chain of
// WAVEFRONT rounds - march kernel -> sample buffers -> shade kernel -> compaction, 4 rounds, the rays still alive left to a
// "straggler" kernel, which was this persistent loop - as two chains over the halves of the rays on two HIP streams, so
// that one half's latency-bound march ran beside the other's gather-bound shade kernel.  That overlap was HIP's to grant
// (how streams are dealt to hardware queues): the same command ran at 396 and 544 frames/s on two boxes of one pool.  Round 6
// wrote the overlap down - one launch carrying the shade stage of one half and the march stage of the other - and
// measured that it is not an overlap at all: such a launch takes the SUM of its two stages' times (a marching workgroup
// holds registers and issue slots like any other), the two-stream chains had merely hidden launch ramps and tails; and
// hipExtAnyOrderLaunch is ignored on gfx9.  On ONE stream the persistent loop alone beats every rounds configuration
// (render of the benchmark view: 0.68 ms against 0.82 for 4 rounds, 0.75 for 1; a frame's Depth + Shade pair 1.19 against
// 1.39): no sample buffers (160 MB written and read back per round), no per-round state traffic, no compaction, 3
// launches instead of 12 - and nothing left for a queue assignment to decide.  The rounds were removed.
//
// Every ray performs exactly the arithmetic of oracle/ngp_oracle.py on exactly the same samples; samples a step evaluates
// past a ray's termination point are discarded.
// ===========================================================================
constexpr int kK = 8;          // samples per ray per step
constexpr int kCtrWords = 16;  // ints per counter block (one 64-B line per pipe)


struct Ray {
  float o[3], d[3], idir[3];
  float tmin, tmax, zdot;
  bool hit;
};

// Slab test against the render box clipped to the scene box.
__device__ inline void ray_clip(const NgpParams& P, Ray& r) {
  const float half_s = P.aabb_scale * 0.5f;
  r.tmin = -INFINITY;
  r.tmax = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = fmaxf(P.lo[a], 0.5f - half_s), hi = fminf(P.hi[a], 0.5f + half_s);
    const float t0 = (lo - r.o[a]) * r.idir[a], t1 = (hi - r.o[a]) * r.idir[a];
    r.tmin = fmaxf(r.tmin, fminf(t0, t1));
    r.tmax = fminf(r.tmax, fmaxf(t0, t1));
  }
  r.hit = r.tmax > fmaxf(r.tmin, 0.f);
}

// The camera: a kernel argument, or - for a render enqueued before its pose is known on the host - 12 floats in
// device memory that a one-thread kernel derived from the pose record of the LM kernel ahead of it in the stream.
// The camera's twelve floats are read through a pointer laundered into a VGPR (plain vector loads, no address arithmetic in
// SGPRs).  Round 6: with another HIP stream's convolution kernels running beside a render, single waves of the ray generator
// computed their rays from a slightly different camera (4th digit; the same wave's second computation was right, the
// kernel-argument memory never changed) - 1 render in 6 beside UNet passes.  The build with this form of the loads is clean
// over 22,000 such renders; the mechanism is NOT established (the defect moves with the instruction stream around the
// loads: profiles/r06_experiments.md section 8), tests/test_edge_cases_gpu.py holds the regression test.
__device__ __forceinline__ const float* camera_pointer(const NgpParams& P) {
  const float* c = P.cam_dev ? P.cam_dev : P.cam;
  asm volatile("" : "+v"(c));
  return c;
}
__device__ inline void load_camera(const NgpParams& P, float* cam) {
  const float* c = camera_pointer(P);
#pragma unroll
  for (int i = 0; i < 12; ++i) cam[i] = c[i];
}

__device__ inline Ray make_ray(const NgpParams& P, int px, int py) {
  Ray r;
  const float u = ((float)px + 0.5f) / (float)P.W, vv = ((float)py + 0.5f) / (float)P.H;
  float dxn = (u - 0.5f) * (float)P.W / P.focal, dyn = (vv - 0.5f) * (float)P.H / P.focal;
  if (P.k1 != 0.f) {
    float xu = dxn, yu = dyn;
    for (int it = 0; it < 8; ++it) {
      const float r2 = xu * xu + yu * yu;
      const float s = 1.0f + P.k1 * r2;
      xu = dxn / s;
      yu = dyn / s;
    }
    dxn = xu;
    dyn = yu;
  }
  float cam[12];
  load_camera(P, cam);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    r.d[a] = (dxn * cam[4 * a + 0] + dyn * cam[4 * a + 1]) + cam[4 * a + 2];
    r.o[a] = cam[4 * a + 3];
  }
  const float nrm = sqrtf((r.d[0] * r.d[0] + r.d[1] * r.d[1]) + r.d[2] * r.d[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    r.d[a] = r.d[a] / nrm;
    r.idir[a] = 1.0f / r.d[a];
  }
  const float fn = sqrtf((cam[2] * cam[2] + cam[6] * cam[6]) + cam[10] * cam[10]);
  r.zdot = (r.d[0] * (cam[2] / fn) + r.d[1] * (cam[6] / fn)) + r.d[2] * (cam[10] / fn);
  ray_clip(P, r);
  return r;
}

// The ray of a stored (unit direction, zdot) record: bit for bit what make_ray returned when
// the record was written (same divisions, same slab arithmetic), without the pixel -> direction
// part.
__device__ inline Ray ray_from_record(const NgpParams& P, const float4 rd) {
  Ray r;
  r.d[0] = rd.x; r.d[1] = rd.y; r.d[2] = rd.z;
  r.zdot = rd.w;
  const float* c = camera_pointer(P);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    r.o[a] = c[4 * a + 3];
    r.idir[a] = 1.0f / r.d[a];
  }
  ray_clip(P, r);
  return r;
}

// One lattice point of a ray: its position, step and cascade, and whether its occupancy cell is set.
__device__ inline bool probe_cell(const NgpParams& P, const Ray& r, float t, float* pos, float& dt, int& mip) {
#pragma unroll
  for (int a = 0; a < 3; ++a) pos[a] = r.o[a] + t * r.d[a];
  dt = calc_dt(t, P.cone_angle, P.dt_lo, P.dt_hi);
  // instant-ngp mip_from_dt: `dt *= 2 * NERF_GRIDSIZE()` (rounds 1-4 had the factor without the 2: one cascade finer
  // wherever t >= 1 - VERDICT r4; the `dt < 1 -> mip_from_pos` guard upstream is what the max() below does for e <= 0)
  const int e = frexp_exp(dt * (float)(2 * kGrid));
  mip = min(P.cascades - 1, max(e, mip_from_pos(pos[0], pos[1], pos[2], P.cascades)));
  const float msc = pow2i(-mip);
  int ci[3];
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = (pos[a] - 0.5f) * msc + 0.5f;
    ci[a] = (int)floorf(p * (float)kGrid);
    inside = inside && ci[a] >= 0 && ci[a] < kGrid;
    ci[a] = min(max(ci[a], 0), kGrid - 1);
  }
  const unsigned lin = (unsigned)((ci[2] * kGrid + ci[1]) * kGrid + ci[0]) +
                       (unsigned)mip * (unsigned)(kGrid * kGrid * kGrid);
  return inside && ((P.occ[lin >> 3] >> (lin & 7u)) & 1u);
}

// advance_to_next_voxel's target: where the ray leaves the (empty) cell of cascade `mip` that holds `pos`
__device__ inline float cell_exit_t(const Ray& r, float t, const float* pos, int mip) {
  const float res = pow2i(7 - mip), ires = pow2i(mip - 7);  // 128 / 2^mip cells per unit
  float tm = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = res * (pos[a] - 0.5f);
    const float sg = r.d[a] > 0.f ? 1.f : (r.d[a] < 0.f ? -1.f : 0.f);
    const float tx = (floorf(p + 0.5f + 0.5f * sg) - p) * r.idir[a];
    if (r.d[a] != 0.f) tm = fminf(tm, tx);
  }
  return t + fmaxf(tm * ires, 0.f);  // exact: res is a power of two
}

// advance_to_next_voxel: step in dt increments past the border of the (empty) cell at `pos`
__device__ inline void advance_past_cell(const NgpParams& P, const Ray& r, float& t, const float* pos, int mip) {
  const float t_target = cell_exit_t(r, t, pos, mip);
  do {
    t = t + calc_dt(t, P.cone_angle, P.dt_lo, P.dt_hi);
  } while (t < t_target);
}

// Advances t along the dt lattice to the next sample whose occupancy cell is set.
// Returns false when the ray leaves the render box first.  pos/dt describe the sample.
__device__ inline bool next_sample(const NgpParams& P, const Ray& r, float& t, float* pos, float& dt) {
  for (;;) {
    if (t >= r.tmax) return false;
    int mip;
    if (probe_cell(P, r, t, pos, dt, mip)) return true;
    advance_past_cell(P, r, t, pos, mip);
  }
}

struct NgpWork {       // one pipe: a slice of a render's ray enumeration
  unsigned* rid;       // [slot] pixel * spp + spp_index of the slot's ray (the compact list raygen writes)
  float* t0;           // [slot] its first lattice position
  float4* dir;         // [slot] (unit direction, d . camera z)
  int* counters;       // [0]: rays in the list
  float4* sppbuf;      // [pixel][spp] finished rays (shared by the pipes of a render)
  float* sppbuf_d;     // mode 2: finished rays' depth
};

__device__ inline void sh_fragments(const float* d, unsigned* shB0, unsigned* shB1) {
  float sh[16];
  sh4_eval(d[0], d[1], d[2], sh);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    shB0[i] = pack_h2(sh[2 * i], sh[2 * i + 1]);
    shB1[i] = pack_h2(sh[8 + 2 * i], sh[8 + 2 * i + 1]);
    swap32(shB0[i], shB1[i]);
  }
}

// Ray id = pixel * spp + sample.  Rays are enumerated sample-fastest over 4x2 pixel blocks,
// so one wave holds the 8 spp passes of 8 neighbouring pixels: at the coarse and middle
// hash levels those 64 samples share grid cells and their gathers coalesce in the L1.
// The start jitter of (pixel, pass) as a fraction of a step.
__device__ inline float ray_jitter(int pix, int s) {
  unsigned h = (unsigned)pix * 747796405u + (unsigned)s * 2891336453u + 1u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

__device__ inline bool enum_ray(const NgpParams& P, long long i, int& px, int& py, int& s) {
  const int bx = (P.W + 3) / 4;
  s = (int)(i % P.spp);
  const long long q = i / P.spp;
  const int blk = (int)(q / 8), within = (int)(q % 8);
  px = (blk % bx) * 4 + (within & 3);
  py = (blk / bx) * 2 + (within >> 2);
  if (!(px < P.W && py < P.H)) return false;
  return true;
}

__device__ inline long long enum_total(const NgpParams& P) {
  return (long long)((P.W + 3) / 4) * ((P.H + 1) / 2) * 8 * P.spp;
}

// Start of a ray (pixel, spp pass): the box entry plus the per-(pixel, pass) jitter, or < 0 when
// the pixel's ray misses the render box.
__device__ inline float ray_start(const NgpParams& P, const Ray& r, int pix, int s) {
  if (!r.hit) return -1.f;
  const float t = fmaxf(r.tmin, 0.f) + 1e-6f;
  return t + ray_jitter(pix, s) * calc_dt(t, P.cone_angle, P.dt_lo, P.dt_hi);
}

// Ray generation + order-preserving compaction of 2048-ray tiles: ONE global atomic per tile (a single
// counter word sustains only ~90 atomics/us, MI355X_MICROARCH.md "dequeue").  Tiles land in the order of their atomics,
// i.e. roughly in dispatch order: neighbours in the list are neighbours in the image.
// (blk / nblk: this workgroup's index among the workgroups working on THIS pipe's slice.)
constexpr int kTile = 2048;
__device__ __forceinline__ void ngp_raygen_body(const NgpParams& P, const NgpWork& Wk, int blk, int nblk) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const long long n = P.enum_hi - P.enum_lo;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long tiles = (n + kTile - 1) / kTile;
  for (long long tile = blk; tile < tiles; tile += nblk) {
    const long long i0 = tile * kTile + (long long)threadIdx.x * 8;
    bool k[8];
    int cnt = 0;
    // A thread's 8 consecutive rays are the passes of one pixel when spp = 8: one make_ray for all.
    float t_start[8];
    unsigned rid_new[8];
    int last_pix = -1;
    Ray r;
    r.hit = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long i = i0 + j;
      int px, py, sp;
      k[j] = false;
      t_start[j] = -1.f;
      rid_new[j] = 0u;
      if (i < n && enum_ray(P, P.enum_lo + i, px, py, sp)) {
        const int pix = py * P.W + px;
        if (pix != last_pix) {
          r = make_ray(P, px, py);
          last_pix = pix;
        }
        t_start[j] = ray_start(P, r, pix, sp);
        rid_new[j] = (unsigned)pix * (unsigned)P.spp + (unsigned)sp;
        k[j] = t_start[j] >= 0.f;
      }
      cnt += k[j] ? 1 : 0;
    }
    int inc = cnt;  // inclusive scan within the wave
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const int o = __shfl_up(inc, m, 64);
      if (lane >= m) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int wave_off = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) wave_off += s_wave[w];
      tile_total += s_wave[w];
    }
    if (threadIdx.x == 0) s_base = tile_total ? atomicAdd(Wk.counters, tile_total) : 0;
    __syncthreads();
    int dst = s_base + wave_off + inc - cnt;
    // the render kernel runs 8 lanes per ray: it reads the direction instead of redoing make_ray's fourteen divisions
    const float4 rdir = make_float4(r.d[0], r.d[1], r.d[2], r.zdot);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!k[j]) continue;
      Wk.rid[dst] = rid_new[j];
      Wk.t0[dst] = t_start[j];
      // (a thread's kept rays share one pixel unless spp is not 8: then the ray is rebuilt for the pixel at hand)
      if (P.spp == 8) {
        Wk.dir[dst] = rdir;
      } else {
        const int pix = (int)(rid_new[j] / (unsigned)P.spp);
        const Ray rr = make_ray(P, pix % P.W, pix / P.W);
        Wk.dir[dst] = make_float4(rr.d[0], rr.d[1], rr.d[2], rr.zdot);
      }
      ++dst;
    }
    __syncthreads();
  }
}

// The next K = 8 samples of the wave's 8 rays, marched by all 64 lanes.  The 8 lanes of a ray probe 8 consecutive
// lattice points per trip - one dependent occupancy load per 8 points instead of one per point - and then every lane
// replays ngp_march_kernel's walk over the 8 results (sample / skip to the border of the empty cell / leave the box), so
// the samples are the serial loop's bit for bit: the lattice t' = t + dt(t) does not depend on what the cells hold.
// Lane k of a ray returns the ray's k-th sample (dt = 0: none), `t` moves on, `exhausted` = the ray left the box.
__device__ __forceinline__ void ngp_march_group(const NgpParams& P, const Ray& r, bool alive, float& t, float4& sp,
                                                float& sample_t, bool& exhausted) {
  const int lane = threadIdx.x & 63, j = lane & 7;
  int k = 0, have = 0;
  bool out = false, busy = alive;
  float pending = -INFINITY;  // a skip that ran past the trip's last point
  float ts = 0.f;
  while (__any(busy)) {
    if (busy) {
      float tt[9];
      tt[0] = t;
#pragma unroll
      for (int i = 0; i < 8; ++i) tt[i + 1] = tt[i] + calc_dt(tt[i], P.cone_angle, P.dt_lo, P.dt_hi);
      float tj = tt[0];
      unsigned ge_tmax = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tj = (j == i) ? tt[i] : tj;
        ge_tmax |= (tt[i] >= r.tmax ? 1u : 0u) << i;
      }
      bool occ = false;
      float target = -INFINITY;
      if (tj < r.tmax && tj >= pending) {  // (the walk cannot reach the other points)
        float pos[3], dt;
        int mip;
        occ = probe_cell(P, r, tj, pos, dt, mip);
        if (!occ) target = cell_exit_t(r, tj, pos, mip);
      }
      // an empty point's successor: the first later point at or past the cell's border (advance_past_cell steps at
      // least once); 8 = beyond this trip
      int nxt = 8;
#pragma unroll
      for (int i = 7; i >= 1; --i) nxt = (i > j && tt[i] >= target) ? i : nxt;
      unsigned word = (occ ? 8u : (unsigned)(nxt - 1)) << (4 * j);
      word |= __shfl_xor(word, 1, 8);
      word |= __shfl_xor(word, 2, 8);
      word |= __shfl_xor(word, 4, 8);
      // The common trip - a ray inside the object: a fresh step, all 8 lattice points occupied and inside the box - has a
      // known outcome (lane j takes point j, the ray moves on to the 9th point), and the passes of a pixel share that
      // fate: when EVERY marching lane of the wave is in it, the general walk below (~150 VALU instructions of a kernel
      // that is VALU-bound) is skipped.  The same values, by construction.
      if (__all(!busy || (k == 0 && word == 0x88888888u && ge_tmax == 0u && pending == -INFINITY))) {
        if (busy) {
          ts = tj;
          have = 1;
          k = kK;
          t = tt[8];
          busy = false;
        }
        continue;
      }
      int cur = 8;
#pragma unroll
      for (int i = 7; i >= 0; --i) cur = (tt[i] >= pending) ? i : cur;
      int last_skip = -1, mine = -1;
      while (cur < 8 && k < kK) {
        if ((ge_tmax >> cur) & 1u) { out = true; break; }
        const unsigned nb = (word >> (4 * cur)) & 15u;
        if (nb & 8u) {
          if (k == j) mine = cur;
          ++k; ++cur; last_skip = -1;
        } else {
          last_skip = cur;
          cur = (int)(nb & 7u) + 1;
        }
      }
      float t_cur = tt[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        t_cur = (cur == i) ? tt[i] : t_cur;
        if (mine == i) { ts = tt[i]; have = 1; }
      }
      const float tg = __shfl(target, (lane & ~7) | (last_skip >= 0 ? last_skip : j), 64);
      // A skip that runs past this trip's last point stays pending - also across a trip none of whose points reaches
      // it (a mip >= 1 cell is 9-32 minimum steps wide: `cur` starts at 8, the walk does not run, and the skip must
      // not be dropped and re-derived from a point inside the same cell).
      if (last_skip >= 0) pending = (tt[8] < tg) ? tg : -INFINITY;
      else if (!(pending > tt[8])) pending = -INFINITY;
      t = t_cur;  // k == 8: the point after the last sample; otherwise the next trip's first point
      busy = !out && k < kK;
    }
  }
  sp = make_float4(0.f, 0.f, 0.f, 0.f);
  sample_t = 0.f;
  if (have) {
    sp.x = r.o[0] + ts * r.d[0]; sp.y = r.o[1] + ts * r.d[1]; sp.z = r.o[2] + ts * r.d[2];
    sp.w = calc_dt(ts, P.cone_angle, P.dt_lo, P.dt_hi);
    sample_t = ts;
  }
  exhausted = out;
}

// min / max over the 64 lanes of a wave, returned to every lane: four DPP steps inside the 16-lane rows (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror), two row broadcasts, one readlane.
template <bool IS_MAX>
__device__ __forceinline__ float wave_minmax(float v) {
  auto step = [&](int ctrl, int row_mask) {
    const int iv = __builtin_bit_cast(int, v);
    int o;
    if (ctrl == 0) o = __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if (ctrl == 1) o = __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if (ctrl == 2) o = __builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false);  // row_half_mirror
    else if (ctrl == 3) o = __builtin_amdgcn_update_dpp(iv, iv, 0x140, 0xF, 0xF, false);  // row_mirror
    else if (ctrl == 4) o = __builtin_amdgcn_update_dpp(iv, iv, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1, 3
    else o = __builtin_amdgcn_update_dpp(iv, iv, 0x143, 0xC, 0xF, false);                 // row_bcast31 -> rows 2, 3
    const float f = __builtin_bit_cast(float, o);
    v = IS_MAX ? fmaxf(v, f) : fminf(v, f);
    (void)row_mask;
  };
  step(0, 0); step(1, 0); step(2, 0); step(3, 0); step(4, 0); step(5, 0);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- wave-cooperative corner fetch (round 6; north_star's "LDS-staged hash-grid lookups" at wave granularity).
// The 64 samples of a wave step usually lie on ONE line - the 8 passes of a pixel share their ray, and 8 consecutive steps of
// it span a stretch of ~0.015 of the unit cube - so at a coarse level all of them sit in a few neighbouring cells, and the
// 64 x 8 corner gathers of the level fetch the same dozen entries over and over: 8 gather instructions at ~16 clocks of the
// CU's address unit each, which is what binds this kernel.  When the stretch spans at most 3 cells per axis
// of a level, the wave fetches the 4 x 4 x 4 lattice box around it with ONE gather instruction (lane = lattice point, dense
// index or hash as the level demands) straight into a 256-byte LDS patch (buffer_load ... lds: no VGPR, no wait until
// use), and every sample reads its 8 corners from the patch with ds_read_b32 at constant offsets.  Same entries, same
// trilinear arithmetic: every feature keeps its bits.
// The lattice point of this lane in the box whose low corner is (lox, loy, loz), as a byte offset into the level's table.
__device__ __forceinline__ int ngp_patch_point_offset(const NgpLevel& Lv, int lane, unsigned lox, unsigned loy, unsigned loz) {
  const unsigned lx = lox + (unsigned)(lane & 3), ly = loy + (unsigned)((lane >> 2) & 3), lz = loz + (unsigned)(lane >> 4);
  unsigned idx;
  if (Lv.hashed) idx = (lx ^ (ly * 2654435761u) ^ (lz * 805459861u)) & (Lv.size - 1u);
  else idx = min(lx + ly * Lv.res + lz * Lv.res * Lv.res, Lv.size - 1u);
  return (int)(idx << 2);
}
// One level's features of this lane's sample from the wave's patch (the arithmetic of ngp_encode_level_uniform).
__device__ __forceinline__ unsigned ngp_encode_level_patch(const unsigned* patch, const NgpLevel& Lv, float ux, float uy, float uz,
                                                           unsigned lox, unsigned loy, unsigned loz, bool in_box) {
#pragma clang fp contract(fast)  // (as ngp_encode_level_uniform)
  const float qx = ux * Lv.scale + 0.5f, qy = uy * Lv.scale + 0.5f, qz = uz * Lv.scale + 0.5f;
  const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
  const float ax = qx - fx, ay = qy - fy, az = qz - fz;
  const unsigned gx = (unsigned)(int)fx, gy = (unsigned)(int)fy, gz = (unsigned)(int)fz;
  // (a lane without a sample holds position 0 - outside the box: it reads the box's first cell, its result is masked)
  const unsigned local = in_box ? (gx - lox) + 4u * (gy - loy) + 16u * (gz - loz) : 0u;
  const unsigned* c0 = patch + local;
  const unsigned vals[8] = {c0[0], c0[1], c0[4], c0[5], c0[16], c0[17], c0[20], c0[21]};
  return ngp_trilinear_pk(vals, ax, ay, az);
}

// The render kernel's body: PERSISTENT waves over the ray list of one pipe.
//
// A wave holds 8 rays (ray position r = lane >> 3, its 8 lanes k = lane & 7) and repeats
//   march   ngp_march_group: every ray's next K = 8 occupied lattice samples, lane k ends up with the ray's k-th sample;
//   shade   the wave's 64 samples: hash-grid gathers (all levels) -> both MLPs on MFMA -> in-order compositing of each
//           ray's 8 samples (8 consecutive lanes), early termination;
//   refill  a ray position whose ray terminated or left the box takes the next ray of the wave's share of the list.
// The gathers of one wave overlap the matrix work and the marching of the others (4 waves per SIMD).  The level loop stays
// rolled (unrolled, all 128 gathers are hoisted: 256 VGPRs, one wave per SIMD) and runs in two halves of 8 levels through an
// 8-KB LDS staging area, which keeps Flo / Fhi on static register indices.  What a ray carries from step to step -
// position t, transmittance, premultiplied colour (+ depth) - lives in registers (replicated in its 8 lanes), its id,
// direction and SH fragments are loaded / formed once per ray: a step's only memory traffic is the occupancy probes and the
// table gathers.  A ray's result depends neither on which rays share its wave nor on the grid.
//
// A wave's rays: groups wave_id, wave_id + n_waves, ... of 8 consecutive list entries (the 8 passes of a pixel: 64 samples
// on one line), handed to the wave's 8 ray positions in that order; a position whose ray has finished takes the next
// one, so that the wave keeps shading 64 samples per step until its share of the list runs out.  (Shares drawn from a shared
// counter instead, 64 entries per atomic: 0.85-1.4 ms per render against 0.63-0.70 - returning atomics on one word are served
// one per ~100 ns, and the 4096 waves' first draw alone takes 0.4 ms.  The grid is larger than what is resident instead: the
// dispatcher is the queue.  A wave's share as CONSECUTIVE groups of the list - neighbouring pixels, whose lines run through the
// same coarse cells - loses to it: 0.70-0.80 ms against 0.65 with 2048 ... 16384 workgroups; the cost of a ray varies along
// the list and the interleaved shares even it out.)
template <int MODE>  // 0 colour, 1 depth, 2 colour AND depth of the same rays in one march
__device__ __forceinline__ void ngp_render_body(const NgpParams& P, const NgpWork& Wk, int rays_per_wg, int blk, int nblk,
                                                half8* s_w, unsigned* s_feat, float* s_rays) {
  const int n = Wk.counters[0];
  if (P.stats && blk == 0 && threadIdx.x == 0) atomicAdd(P.stats + 1, (unsigned long long)n);  // rays that hit the box
  // Workgroups that take part.  Up to the knee (4096 workgroups: the optimum of the 640 x 480 x 8 renders, section 1 of
  // profiles/r06_experiments.md) one per `rays_per_wg` rays, at least a quarter of the knee (a short list spread over few
  // waves is a long chain of steps per wave); lists too long for the knee - 1920 x 1080 and 2016 x 1512 renders - get one
  // workgroup per 21 / 16 x rays_per_wg rays (84 at the default 64) up to the launched grid: +6 % on those renders.
  const int knee = min(nblk, 4096);
  const int by_div = (n + rays_per_wg - 1) / rays_per_wg;
  const int by_div2 = (int)(((long long)n * 16) / ((long long)rays_per_wg * 21));
  const int n_wg = min(nblk, max(knee / 4, max(min(by_div, knee), by_div2)));
  if (blk >= n_wg || blk * 32 >= n) return;  // (workgroup-uniform)
  for (int i = threadIdx.x; i < kNumFrags * 64; i += 256) s_w[i] = P.wfrag[i];
  const __amdgpu_buffer_rsrc_t grid_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.grid, 0, (int)P.grid_bytes, 0x00020000);
  const float enc_lo = 0.5f - P.aabb_scale * 0.5f, enc_inv = 1.0f / P.aabb_scale;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rlane = lane >> 3, k = lane & 7;
  unsigned long long n_samples = 0;
  int n_steps = 0, n_box_levels = 0;  // (wave-uniform) shade steps of this wave, levels it fetched as a box
  const int n_waves = n_wg * 4, wave_id = blk * 4 + wave;
  auto stream_slot = [&](int s) { return ((s >> 3) * n_waves + wave_id) * 8 + (s & 7); };
  // ---- the lane's ray (the same in the 8 lanes of a ray position)
  int cursor = 8;
  int slot = stream_slot(rlane);
  bool alive = slot < n;
  // What a ray accumulates - transmittance, premultiplied colour, depth - and its id are only touched between the MLPs and
  // the next march: they live in LDS (8 dwords per ray position; the 8 lanes of a ray read one address), not in registers
  // that would be dead weight across the gathers (the kernel sits at the 128-VGPR line of 4 waves per SIMD).
  float* const s_ray = s_rays + (wave * 8 + rlane) * 8;  // [T, acc.x, acc.y, acc.z, acc.w, depth, ray id, -]
  float t = 0.f;
  float4 rd = make_float4(0.f, 0.f, 1.f, 0.f);
  if (alive) {
    t = Wk.t0[slot];
    rd = Wk.dir[slot];
  }
  if (k == 0) {
    s_ray[0] = 1.f;
    s_ray[1] = s_ray[2] = s_ray[3] = s_ray[4] = s_ray[5] = 0.f;
    s_ray[6] = __builtin_bit_cast(float, alive ? Wk.rid[slot] : 0u);
  }
  unsigned* const col = s_feat + wave * (8 * 64) + lane;  // the features pass through the lane's own LDS column
  // gather in ray-fastest lane order: lane 8 a + b fetches the sample of lane 8 b + a, so that adjacent lanes hold
  // the same step of neighbouring rays (the passes of one pixel: positions a fraction of a step apart on one line)
  // instead of consecutive steps of one ray: the address unit merges the lanes of a quad that share a line.
  // Render 0.713 -> 0.676 ms.  (Rays ranked by distance within a step on top of that: 0.688, the ranking costs more.)
  const int tl = ((lane & 7) << 3) | (lane >> 3);
  unsigned* const wcol = s_feat + wave * (8 * 64) + tl;
  while (__any(alive)) {
    // ---- march: the ray's next 8 samples; lane k gets the k-th
    float4 sp;
    float sample_t;
    bool exhausted;
    {
      const Ray r = ray_from_record(P, rd);
      ngp_march_group(P, r, alive, t, sp, sample_t, exhausted);
    }
    // ---- shade
    const float dt = alive ? sp.w : 0.f;
    const bool valid = dt != 0.f;
    unsigned Flo[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, Fhi[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    float logit = 0.f, rgbv[3] = {0.f, 0.f, 0.f};
    // rays that crossed the box without meeting an occupied cell arrive with eight empty slots,
    // and neighbouring rays share that fate: whole waves skip the MLPs (wave-uniform branch)
    if (__any(valid)) {
      // every lane gathers (empty slots hold position 0: in range), the result is masked afterwards
      float ux = (sp.x - enc_lo) * enc_inv, uy = (sp.y - enc_lo) * enc_inv, uz = (sp.z - enc_lo) * enc_inv;
      ux = __shfl(ux, tl, 64); uy = __shfl(uy, tl, 64); uz = __shfl(uz, tl, 64);
      const unsigned long long vmask = __ballot(valid);
      const bool valid_t = (vmask >> tl) & 1ull;  // (of the sample this lane gathers for)
      ++n_steps;
      // ---- which levels the wave can fetch cooperatively: the lanes with a sample share ONE ray (the passes of a pixel),
      // and the stretch between its first and last sample of this step spans < 2 cells of the level per axis (then at most
      // 3 cells, 4 lattice points).  The stretch's ends are two of the samples themselves, formed with the march's own
      // arithmetic, and cell coordinates are monotone along the line: the box of the ends' cells holds every sample's.
      float coop_scale = 0.f;          // levels with scale <= coop_scale are fetched as a box
      float emin[3] = {0.f, 0.f, 0.f}; // the stretch's low corner in encoder coordinates
      if (P.coop) {
        const int first = __ffsll((long long)vmask) - 1;
        const float d0[3] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rd.x), first)),
                             __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rd.y), first)),
                             __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rd.z), first))};
        if (__all(!valid || (rd.x == d0[0] && rd.y == d0[1] && rd.z == d0[2]))) {
          const float tlo = wave_minmax<false>(valid ? sample_t : INFINITY), thi = wave_minmax<true>(valid ? sample_t : -INFINITY);
          float ext = 0.f;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float o = camera_pointer(P)[4 * a + 3];
            const float ea = ((o + tlo * d0[a]) - enc_lo) * enc_inv, eb = ((o + thi * d0[a]) - enc_lo) * enc_inv;
            // (wave-uniform values: kept in scalar registers)
            emin[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fminf(ea, eb))));
            ext = fmaxf(ext, fabsf(eb - ea));
          }
          // scale x extent <= 1.98: the cell coordinates of the two ends differ by at most 2 (their fp32 error is < 0.002
          // up to the finest level's 8192 cells)
          coop_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.98f / ext)));
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int l0 = 8 * half;
        int n_coop = 0;
        if (P.coop) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the rows' last readers are done before a fetch lands in them)
          while (n_coop < 8 && P.lv[l0 + n_coop].scale <= coop_scale) {
            const NgpLevel& Lv = P.lv[l0 + n_coop];
            unsigned lo[3];
            {
#pragma clang fp contract(fast)  // (the lanes' own q = u * scale + 0.5 is contracted: ngp_encode_level_uniform)
              lo[0] = (unsigned)(int)floorf(emin[0] * Lv.scale + 0.5f);
              lo[1] = (unsigned)(int)floorf(emin[1] * Lv.scale + 0.5f);
              lo[2] = (unsigned)(int)floorf(emin[2] * Lv.scale + 0.5f);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(grid_rsrc, (__attribute__((address_space(3))) void*)(s_feat + wave * (8 * 64) + n_coop * 64),
                                                     4, ngp_patch_point_offset(Lv, lane, lo[0], lo[1], lo[2]), (int)(Lv.offset * 4u), 0, 0);
            ++n_coop;
          }
          n_box_levels += n_coop;
        }
        // the other levels: 8 corner gathers per lane
#pragma unroll 4
        for (int l = n_coop; l < 8; ++l) wcol[l * 64] = ngp_encode_level_uniform(grid_rsrc, P.lv[l0 + l], ux, uy, uz);
        if (n_coop) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the patches have landed
          __builtin_amdgcn_wave_barrier();
          for (int l = 0; l < n_coop; ++l) {
            const NgpLevel& Lv = P.lv[l0 + l];
            unsigned lo[3];
            {
#pragma clang fp contract(fast)
              lo[0] = (unsigned)(int)floorf(emin[0] * Lv.scale + 0.5f);
              lo[1] = (unsigned)(int)floorf(emin[1] * Lv.scale + 0.5f);
              lo[2] = (unsigned)(int)floorf(emin[2] * Lv.scale + 0.5f);
            }
            const unsigned pk = ngp_encode_level_patch(s_feat + wave * (8 * 64) + l * 64, Lv, ux, uy, uz, lo[0], lo[1], lo[2], valid_t);
            __builtin_amdgcn_wave_barrier();  // every lane has read the patch before a feature replaces part of it
            wcol[l * 64] = pk;
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (half == 0) {
#pragma unroll
          for (int l = 0; l < 8; ++l) Flo[l] = valid ? col[l * 64] : 0u;
        } else {
#pragma unroll
          for (int l = 0; l < 8; ++l) Fhi[l] = valid ? col[l * 64] : 0u;
        }
        __builtin_amdgcn_wave_barrier();  // (the next writers of the rows must not overtake these reads)
      }
      // (the SH fragments of the ray's direction are formed here, per step, instead of living in 8 registers across the
      // march and the gathers: ~60 VALU instructions of a step's ~2500)
      unsigned shB0[4], shB1[4];
      const float rdir[3] = {rd.x, rd.y, rd.z};
      sh_fragments(rdir, shB0, shB1);
      ngp_mlp<MODE == 1>(s_w, lane, Flo, Fhi, shB0, shB1, logit, rgbv);
    }
    // ---- in-order compositing of the ray's K samples (8 consecutive lanes) ----
    float alpha = 0.f;
    if (valid) alpha = 1.0f - expf(-expf(logit) * dt);
    float depth = 0.f;
    if (MODE != 0) depth = (sample_t * rd.w) * P.depth_scale;
    if (MODE == 1) rgbv[0] = rgbv[1] = rgbv[2] = depth;
    // inclusive product scan of (1 - alpha) over the 8 lanes of the ray
    float pinc = 1.0f - alpha;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      const float o = __shfl_up(pinc, m, 8);
      if (k >= m) pinc = pinc * o;
    }
    float pexc = __shfl_up(pinc, 1, 8);
    if (k == 0) pexc = 1.0f;
    const float T0 = s_ray[0];
    const float T_before = T0 * pexc, T_after = T0 * pinc;
    // the first sample after which T drops below the threshold ends the ray (it is included)
    const bool ends = valid && (T_after < P.min_T);
    const unsigned long long bal = __ballot(ends);
    const unsigned grp = (unsigned)((bal >> (rlane * 8)) & 0xFFull);
    const int k_term = grp ? (__ffs((int)grp) - 1) : 8;
    const bool contributes = valid && k <= k_term;
    const float wgt = contributes ? alpha * T_before : 0.f;
    float cr = wgt * rgbv[0], cg = wgt * rgbv[1], cb = wgt * rgbv[2], ca = wgt, cd = wgt * depth;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      cr += __shfl_xor(cr, m, 8);
      cg += __shfl_xor(cg, m, 8);
      cb += __shfl_xor(cb, m, 8);
      ca += __shfl_xor(ca, m, 8);
      if (MODE == 2) cd += __shfl_xor(cd, m, 8);
    }
    if (contributes) n_samples += 1;
    // last valid sample's T_after (or the terminating one) is the ray's new transmittance
    const unsigned long long vb = __ballot(valid);
    const unsigned vgrp = (unsigned)((vb >> (rlane * 8)) & 0xFFull);
    const int n_valid = __popc(vgrp);
    const int k_last = grp ? k_term : n_valid - 1;
    const float T_new = (k_last >= 0) ? __shfl(T_after, rlane * 8 + max(k_last, 0), 64) : T0;
    if (alive) {
      const bool terminated = grp != 0;
      const bool closes = terminated || exhausted;
      if (k == 0) {  // one lane per ray keeps the books
        float4 acc = make_float4(s_ray[1] + cr, s_ray[2] + cg, s_ray[3] + cb, s_ray[4] + ca);
        float accd = 0.f;
        if (MODE == 2) accd = s_ray[5] + cd;
        if (closes) {
          const unsigned rid = __builtin_bit_cast(unsigned, s_ray[6]);
          if (MODE == 2) Wk.sppbuf_d[rid] = terminated ? accd / acc.w : accd;
          if (terminated) {
            acc.x /= acc.w; acc.y /= acc.w; acc.z /= acc.w; acc.w = 1.0f;
          }
          Wk.sppbuf[rid] = acc;
        } else {
          s_ray[0] = T_new;
          s_ray[1] = acc.x; s_ray[2] = acc.y; s_ray[3] = acc.z; s_ray[4] = acc.w;
          if (MODE == 2) s_ray[5] = accd;
        }
      }
      if (closes) alive = false;
    }
    // ---- refill the free positions, lowest first
    const unsigned long long dead = __ballot(!alive && k == 0);  // bit 8 r: ray position r is free
    if (dead) {
      const int rank = __popcll(dead & ((1ull << (rlane * 8)) - 1ull));
      if (!alive) {
        const int ns = stream_slot(cursor + rank);
        if (ns < n) {
          alive = true;
          t = Wk.t0[ns];
          rd = Wk.dir[ns];
          if (k == 0) {
            s_ray[0] = 1.f;
            s_ray[1] = s_ray[2] = s_ray[3] = s_ray[4] = s_ray[5] = 0.f;
            s_ray[6] = __builtin_bit_cast(float, Wk.rid[ns]);
          }
        }
      }
      cursor += __popcll(dead);
    }
  }
  if (P.stats) {  // one atomic per workgroup and counter: a single counter word sustains only ~90 atomics/us
    __shared__ unsigned long long s_cnt[4][3];
    for (int m = 32; m >= 1; m >>= 1) n_samples += __shfl_xor(n_samples, m, 64);
    if (lane == 0) {
      s_cnt[wave][0] = n_samples;
      s_cnt[wave][1] = (unsigned long long)n_steps;
      s_cnt[wave][2] = (unsigned long long)n_box_levels;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      const unsigned long long tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
      // [0] samples composited, [2] wave steps that shaded samples (64 sample slots each), [3] levels of them fetched as a box
      if (tot) atomicAdd(P.stats + (threadIdx.x == 0 ? 0 : threadIdx.x + 1), tot);
    }
  }
}

// The camera of a render whose pose the host has not seen yet: pose_to_camera_f64 (pxt_common.h) in a one-thread launch
// behind the LM kernel (pxt_ngp_render_both_from_pose), or - one dispatch less - in the LM kernel's own epilogue
// (pxt_lm_refine_cam writes this context's camera slot; pxt_ngp_render_frame(camera_from_slot = 1) reads it).
__global__ void ngp_pose_to_camera_kernel(const float* __restrict__ pose12, const PoseConv cv, float* __restrict__ cam_dev,
                                          float* __restrict__ cam_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float cam[12];
  pose_to_camera_f64(pose12, cv, cam);
  for (int i = 0; i < 12; ++i) {
    cam_dev[i] = cam[i];
    if (cam_out) cam_out[i] = cam[i];
  }
  // cam_out[12] flips to 1 once the 12 floats are visible system-wide (the host polls it in pinned memory)
  if (cam_out) __hip_atomic_store(&cam_out[12], 1.f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// instant-ngp's srgb_to_linear (common_device.cuh), applied by its shade_kernel_nerf to every finished ray's
// composited colour when the snapshot was not trained in linear colours.  pow through v_log_f32 / v_exp_f32
// (1 ulp each): relative error < 3e-6 over (0.04045, 1], far inside the renderer's 5e-4 parity bar.
__device__ inline float srgb_to_linear(float c) {
  if (c <= 0.04045f) return c / 12.92f;
  return __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf((c + 0.055f) / 1.055f));
}

// The last kernel of a render also zeroes the round counters of every pipeline for the NEXT render (everything that
// reads them has finished by now): no memset launch in front of a render's first kernel.
struct NgpCounterList { int* p[4]; int n; };
__device__ __forceinline__ void ngp_resolve_body(const NgpParams& P, const NgpWork& Wk, const NgpCounterList& zl, int blk) {
  if (blk == 0)
    for (int w = 0; w < zl.n; ++w)
      for (int i = threadIdx.x; i < kCtrWords; i += 256) zl.p[w][i] = 0;
  // One lane per pixel: it reads the pixel's spp finished rays (contiguous: 16 B x spp, whole lines per lane)
  // and adds them in pass order - the fixed order of a sequential mean.  All passes of a pixel share one ray
  // (snap_to_pixel_centers), so a pixel whose ray misses the box has no finished rays to read: nothing
  // zero-fills the buffers.  (The first version spread a pixel over 8 lanes and funnelled the passes through
  // 40 ds_bpermute shuffles per pixel: 21.6 us per 640x480x8 resolve.)
  const int wh = P.W * P.H;
  const int pix = blk * 256 + threadIdx.x;
  if (pix >= wh) return;
  const bool hit = make_ray(P, pix % P.W, pix / P.W).hit;
  float ar = 0.f, ag = 0.f, ab = 0.f, aa = 0.f, ad = 0.f;
  if (hit) {
    const float4* src = Wk.sppbuf + (size_t)pix * P.spp;
    const float* srcd = Wk.sppbuf_d + (size_t)pix * P.spp;
    for (int s0 = 0; s0 < P.spp; s0 += 8) {
      float4 v[8];
      float vd[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = (s0 + k < P.spp) ? src[s0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        vd[k] = (P.mode == 2 && s0 + k < P.spp) ? srcd[s0 + k] : 0.f;
      }
      if (P.mode != 1 && P.srgb_to_linear) {  // Shade colours only: mode 1's buffer holds depths
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[k].x = srgb_to_linear(v[k].x); v[k].y = srgb_to_linear(v[k].y); v[k].z = srgb_to_linear(v[k].z);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // sequential: ((v0 + v1) + v2) + ...
        if (s0 + k < P.spp) {
          ar += v[k].x; ag += v[k].y; ab += v[k].z; aa += v[k].w;
          if (P.mode == 2) ad += vd[k];
        }
      }
    }
  }
  const float inv = 1.0f / (float)P.spp;
  float depth_x = 0.f;
  if (P.mode == 2) {  // what a separate Depth-mode render would have written
    float4 od;
    od.w = aa * inv;
    od.x = od.y = od.z = ad * inv;
    od.x += P.bg[0] * P.bg[3] * (1.0f - od.w);
    od.y += P.bg[1] * P.bg[3] * (1.0f - od.w);
    od.z += P.bg[2] * P.bg[3] * (1.0f - od.w);
    od.w = od.w + P.bg[3] * (1.0f - od.w);
    if (P.out_depth) *(float4*)(P.out_depth + 4 * (size_t)pix) = od;
    depth_x = od.x;
  }
  float4 o;
  o.w = aa * inv;
  o.x = ar * inv + P.bg[0] * P.bg[3] * (1.0f - o.w);
  o.y = ag * inv + P.bg[1] * P.bg[3] * (1.0f - o.w);
  o.z = ab * inv + P.bg[2] * P.bg[3] * (1.0f - o.w);
  o.w = o.w + P.bg[3] * (1.0f - o.w);
  if (P.out) *(float4*)(P.out + 4 * (size_t)pix) = o;
  if (P.mode == 1) depth_x = o.x;
  // The 8-bit planes the tracker consumes, written here instead of by two more launches over the float images
  // (get_nerf_image's `(rgb * 255).astype(uint8)` with alpha_thresh 0, run_vis_on_poses.py:52-54; get_mask's
  // `uint8(depth * 255) != 0`, pixloc_tracker_r9.py:210-212): same float -> integer truncation, mod 256.
  if (P.out_u8 && P.mode != 1) {
    uint8_t* q = P.out_u8 + 3 * (size_t)pix;
    q[0] = (uint8_t)((long long)(o.x * 255.0f) & 255);
    q[1] = (uint8_t)((long long)(o.y * 255.0f) & 255);
    q[2] = (uint8_t)((long long)(o.z * 255.0f) & 255);
  }
  if (P.out_nz && P.mode != 0) P.out_nz[pix] = (((long long)(depth_x * 255.0f) & 255) != 0) ? 1 : 0;
}

// ---- the launches.  blockIdx.y = pipe: a render is cut into pipes (slices of its ray enumeration, one by default), a chain
// carries the pipes of one render, of a frame's two renders (the mask's Depth at the query camera + the reference image's
// Shade at the reference camera, pixtrack/pose_trackers/pixloc_tracker_r9.py:145-152,207-214) or of K objects tracked in
// lock-step - every launch carries all of them.  The pipes' parameter records travel by value in the kernel-argument segment
// while they fit (<= 4: a render, a frame's pair), else they sit in device memory (uploaded from a pinned ring ahead of the
// chain).  Read-only for the whole chain and addressed uniformly per workgroup: scalar loads either way.
// (By value they MUST be the kernel's first parameter: the kernels address them through the kernel-argument segment
// pointer - indexing the by-value aggregate itself makes the compiler copy all of it to scratch first, 3 KB per lane.  In
// memory the pointer is a __restrict__ kernel parameter, which is what lets the compiler keep the loads scalar.)
constexpr int kMaxChainPipes = 4 * PXT_NGP_MAX_BATCH;  // (pxt_ngp::kMaxPipes per render)
struct NgpBatchItem {
  NgpParams P;
  NgpWork W;
  NgpCounterList zl;
};
template <int NV>
struct NgpItemsByValue {
  NgpBatchItem it[NV];
};
__device__ __forceinline__ const NgpBatchItem& ngp_kernarg_item(int i) {
  return ((const NgpBatchItem*)__builtin_amdgcn_kernarg_segment_ptr())[i];
}

__global__ __launch_bounds__(256) void ngp_raygen_kernel_v(const NgpItemsByValue<4> items) {
  const NgpBatchItem& it = ngp_kernarg_item(blockIdx.y);
  ngp_raygen_body(it.P, it.W, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void ngp_raygen_kernel_m(const NgpBatchItem* __restrict__ items) {
  const NgpBatchItem& it = items[blockIdx.y];
  ngp_raygen_body(it.P, it.W, blockIdx.x, gridDim.x);
}

// MODES: 0 / 1 / 2 = every pipe of the chain renders in that mode; 3 = per pipe (P.mode; a frame's Depth + Shade pair).
template <int MODES>
__device__ __forceinline__ void ngp_render_impl(const NgpBatchItem& it, int rays_per_wg) {
  __shared__ half8 s_w[kNumFrags * 64];
  __shared__ unsigned s_feat[4 * 8 * 64];
  __shared__ float s_rays[4 * 8 * 8];
  const int mode = MODES == 3 ? it.P.mode : MODES;
  if (mode == 0) ngp_render_body<0>(it.P, it.W, rays_per_wg, blockIdx.x, gridDim.x, s_w, s_feat, s_rays);
  else if (mode == 1) ngp_render_body<1>(it.P, it.W, rays_per_wg, blockIdx.x, gridDim.x, s_w, s_feat, s_rays);
  else ngp_render_body<2>(it.P, it.W, rays_per_wg, blockIdx.x, gridDim.x, s_w, s_feat, s_rays);
}
template <int MODES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ngp_render_kernel_v(const NgpItemsByValue<4> items,
                                                                                                   int rays_per_wg) {
  ngp_render_impl<MODES>(ngp_kernarg_item(blockIdx.y), rays_per_wg);
}
template <int MODES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ngp_render_kernel_m(
    const NgpBatchItem* __restrict__ items, int rays_per_wg) {
  ngp_render_impl<MODES>(items[blockIdx.y], rays_per_wg);
}

// blockIdx.y = render (the items here are per RENDER: the whole view, the shared per-ray result buffers, every counter block)
__global__ __launch_bounds__(256) void ngp_resolve_kernel_v(const NgpItemsByValue<4> items) {
  const NgpBatchItem& it = ngp_kernarg_item(blockIdx.y);
  ngp_resolve_body(it.P, it.W, it.zl, blockIdx.x);
}
__global__ __launch_bounds__(256) void ngp_resolve_kernel_m(const NgpBatchItem* __restrict__ items) {
  const NgpBatchItem& it = items[blockIdx.y];
  ngp_resolve_body(it.P, it.W, it.zl, blockIdx.x);
}

// Network query at caller-given points (unit tests / debugging): out[n] = (logit, r, g, b).
__global__ __launch_bounds__(256) void ngp_query_kernel(const NgpParams P, const float* __restrict__ pos,
                                                        const float* __restrict__ dir, int n,
                                                        float* __restrict__ out) {
  __shared__ half8 s_w[kNumFrags * 64];
  for (int i = threadIdx.x; i < kNumFrags * 64; i += 256) s_w[i] = P.wfrag[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < n;
  const int ii = ok ? i : 0;
  const float half_s = P.aabb_scale * 0.5f;
  const float scene_lo = 0.5f - half_s, inv_s = 1.0f / P.aabb_scale;
  float sh[16];
  sh4_eval(dir[3 * ii], dir[3 * ii + 1], dir[3 * ii + 2], sh);
  unsigned shB0[4], shB1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    shB0[k] = pack_h2(sh[2 * k], sh[2 * k + 1]);
    shB1[k] = pack_h2(sh[8 + 2 * k], sh[8 + 2 * k + 1]);
    swap32(shB0[k], shB1[k]);
  }
  float logit, rgbv[3];
  ngp_eval<false>(P, s_w, lane, ok, (pos[3 * ii] - scene_lo) * inv_s, (pos[3 * ii + 1] - scene_lo) * inv_s,
           (pos[3 * ii + 2] - scene_lo) * inv_s, shB0, shB1, logit, rgbv);
  if (ok) {
    out[4 * i] = logit;
    out[4 * i + 1] = rgbv[0];
    out[4 * i + 2] = rgbv[1];
    out[4 * i + 3] = rgbv[2];
  }
}

}  // namespace pxt

// The read-only tables of a snapshot: shared by a context and the contexts made from it with pxt_ngp_create_shared (a frame's
// second render - another view of the SAME NeRF - needs its own ray lists, counters and camera slot, not a second copy of
// 30 MB of tables competing for the same caches).
struct NgpTables {
  unsigned* grid = nullptr;
  unsigned grid_bytes = 0;
  pxt::half8* wfrag = nullptr;
  uint8_t* occ = nullptr;
  int refs = 1;
};

struct pxt_ngp {
  pxt_ngp_model model;
  NgpTables* tab = nullptr;
  float* cam_dev = nullptr;  // 12 floats: the camera of a render enqueued ahead of its pose (render_both_from_pose)
  pxt::NgpLevel lv[pxt::kMaxLevels];
  // scratch of the wavefront renderer, grown on demand (rays = W*H*spp)
  void* scratch = nullptr;
  size_t scratch_rays = 0;
  size_t scratch_cap = 0;   // rays one pipe's buffers hold
  int scratch_pipes = 0;    // pipes the scratch was laid out for
  bool counters_clean = false;  // the previous render's resolve kernel zeroed every pipe's round counters
  static constexpr int kMaxPipes = 4;
  pxt::NgpWork work[kMaxPipes];            // independent pipes over equal slices of the rays
  int pipelines = 0;     // 0: default (PXT_NGP_PIPES or 2); else the number of ray slices a large render is cut into
  int timing = 0;        // > 0: HIP events around the shade-carrying launches of every timing-th render
  long long renders = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // recorded, not yet read
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;    // created once, reused
};

using namespace pxt;

namespace {

// Column permutations that make one layer's D fragment the next layer's B fragment.
// kind 0: raw features (lane-swapped): slot(q,h,j) -> 16h + 8q + j
// kind 1: hidden D layout, K = 64:      slot(q,h,j) -> 32(q>>1) + 16(q&1) + 8(j>>2) + 4h + (j&3)
// kind 2: colour input: q=0 density outputs in D layout, q=1 SH (lane-swapped)
int kappa(int kind, int q, int h, int j) {
  if (kind == 0) return 16 * h + 8 * q + j;
  if (kind == 1) return 32 * (q >> 1) + 16 * (q & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
  if (q == 0) return 8 * (j >> 2) + 4 * h + (j & 3);
  return 16 + 8 * h + j;
}

void pack_layer(const uint16_t* W, int n_out, int n_in, int kind, int frag0, std::vector<uint16_t>& dst) {
  const int n_rb = (n_out + 31) / 32, n_q = n_in / 16;
  for (int rb = 0; rb < n_rb; ++rb)
    for (int q = 0; q < n_q; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int row = 32 * rb + (lane & 31), h = lane >> 5;
          const int col = kappa(kind, q, h, j);
          const uint16_t v = (row < n_out) ? W[(size_t)row * n_in + col] : (uint16_t)0;
          dst[((size_t)(frag0 + rb * n_q + q) * 64 + lane) * 8 + j] = v;
        }
}

void release_tables(NgpTables* t) {
  if (!t || --t->refs > 0) return;
  if (t->grid) (void)hipFree(t->grid);
  if (t->wfrag) (void)hipFree(t->wfrag);
  if (t->occ) (void)hipFree(t->occ);
  delete t;
}

// A context's camera slot holds a valid camera from the start: a render queued with camera_from_slot behind a refinement
// that failed (the LM epilogue then leaves the slot alone, pxt_lm.hip) must march SOMETHING well defined - its frame is
// dropped afterwards, but its kernels run.
int init_camera_slot(pxt_ngp* ctx) {
  const float cam0[16] = {1.f, 0.f, 0.f, 0.5f, 0.f, 1.f, 0.f, 0.5f, 0.f, 0.f, 1.f, -1.f, 0.f, 0.f, 0.f, 0.f};
  hipError_t e = hipMalloc((void**)&ctx->cam_dev, sizeof(cam0));
  if (e == hipSuccess) e = hipMemcpy(ctx->cam_dev, cam0, sizeof(cam0), hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_last_error("pxt_ngp camera slot", e); return PXT_E_HIP; }
  return PXT_OK;
}

}  // namespace

extern "C" int pxt_ngp_create(const pxt_ngp_model* model, const void* grid_params, int64_t n_grid_params,
                              const void* mlp_params, int64_t n_mlp_params, const uint8_t* occupancy,
                              int64_t n_occ_bytes, pxt_ngp** out_ctx) {
  if (!model || !grid_params || !mlp_params || !occupancy || !out_ctx) return PXT_E_ARG;
  if (model->n_levels < 1 || model->n_levels > kMaxLevels || model->n_features != 2) return PXT_E_ARG;
  if (model->n_levels != 16) return PXT_E_ARG;  // the MLP input width is 32 = 16 levels x 2
  if (model->grid_cascades < 1 || model->grid_cascades > 8) return PXT_E_ARG;
  if (n_mlp_params != 64 * 32 + 16 * 64 + 64 * 32 + 64 * 64 + 16 * 64) return PXT_E_ARG;
  if (n_occ_bytes != (int64_t)model->grid_cascades * kGrid * kGrid * kGrid / 8) return PXT_E_ARG;
  pxt_ngp* ctx = new pxt_ngp();
  ctx->model = *model;
  // tiny-cuda-nn GridEncoding level layout
  const unsigned Tsz = 1u << model->log2_hashmap;
  unsigned off = 0;
  for (int l = 0; l < model->n_levels; ++l) {
    const double scale = std::exp2((double)l * std::log2((double)model->per_level_scale)) * model->base_res - 1.0;
    const unsigned res = (unsigned)std::ceil(scale) + 1;
    unsigned long long n = (unsigned long long)res * res * res;
    unsigned sz = n > Tsz ? Tsz : (unsigned)n;
    sz = (sz + 7) / 8 * 8;
    sz = sz < Tsz ? sz : Tsz;
    ctx->lv[l].scale = (float)scale;
    ctx->lv[l].res = res;
    ctx->lv[l].offset = off;
    ctx->lv[l].size = sz;
    ctx->lv[l].hashed = n > sz ? 1u : 0u;
    off += sz;
  }
  if (n_grid_params != (int64_t)off * 2) { delete ctx; return PXT_E_ARG; }
  std::vector<uint16_t> frag((size_t)kNumFrags * 64 * 8);
  const uint16_t* mp = (const uint16_t*)mlp_params;
  pack_layer(mp, 64, 32, 0, kFragD1, frag);
  pack_layer(mp + 2048, 16, 64, 1, kFragD2, frag);
  pack_layer(mp + 3072, 64, 32, 2, kFragC1, frag);
  pack_layer(mp + 5120, 64, 64, 1, kFragC2, frag);
  pack_layer(mp + 9216, 16, 64, 1, kFragC3, frag);
  NgpTables* t = ctx->tab = new NgpTables();
  t->grid_bytes = (unsigned)((size_t)off * 4);
  hipError_t e = hipMalloc((void**)&t->grid, (size_t)off * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&t->wfrag, frag.size() * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&t->occ, (size_t)n_occ_bytes);
  if (e == hipSuccess) e = hipMemcpy(t->grid, grid_params, (size_t)off * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(t->wfrag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(t->occ, occupancy, (size_t)n_occ_bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_last_error("pxt_ngp_create", e);
    pxt_ngp_destroy(ctx);
    return PXT_E_HIP;
  }
  if (const int rc = init_camera_slot(ctx)) { pxt_ngp_destroy(ctx); return rc; }
  *out_ctx = ctx;
  return PXT_OK;
}

extern "C" int pxt_ngp_create_shared(pxt_ngp* src, pxt_ngp** out_ctx) {
  if (!src || !src->tab || !out_ctx) return PXT_E_ARG;
  pxt_ngp* ctx = new pxt_ngp();
  ctx->model = src->model;
  for (int l = 0; l < kMaxLevels; ++l) ctx->lv[l] = src->lv[l];
  ctx->tab = src->tab;
  ++ctx->tab->refs;
  ctx->pipelines = src->pipelines;
  if (const int rc = init_camera_slot(ctx)) { pxt_ngp_destroy(ctx); return rc; }
  *out_ctx = ctx;
  return PXT_OK;
}

extern "C" int pxt_ngp_destroy(pxt_ngp* ctx) {
  if (!ctx) return PXT_E_ARG;
  release_tables(ctx->tab);
  if (ctx->cam_dev) (void)hipFree(ctx->cam_dev);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (auto& ev : ctx->pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  delete ctx;
  return PXT_OK;
}

static void fill_model(const pxt_ngp* ctx, NgpParams& P) {
  std::memset(&P, 0, sizeof(P));
  P.grid = ctx->tab->grid;
  P.grid_bytes = ctx->tab->grid_bytes;
  P.wfrag = ctx->tab->wfrag;
  P.occ = ctx->tab->occ;
  for (int l = 0; l < kMaxLevels; ++l) P.lv[l] = ctx->lv[l < ctx->model.n_levels ? l : 0];
  P.n_levels = ctx->model.n_levels;
  P.cascades = ctx->model.grid_cascades;
  P.aabb_scale = ctx->model.aabb_scale;
  P.cone_angle = ctx->model.cone_angle;
  P.depth_scale = ctx->model.depth_scale;
  P.srgb_to_linear = ctx->model.linear_colors ? 0 : 1;
  static const int coop_rt = [] { const char* e = getenv("PXT_NGP_COOP"); return e ? atoi(e) : 1; }();
  P.coop = coop_rt;
  P.dt_lo = (float)(std::sqrt(3.0) / 1024.0);
  P.dt_hi = P.dt_lo * (float)(1 << (P.cascades - 1)) * (float)(1024 / kGrid);
}

extern "C" int pxt_ngp_query(pxt_ngp* ctx, const float* pos, const float* dir, int32_t n, float* out,
                             void* stream) {
  if (!ctx || !pos || !dir || !out || n < 1) return PXT_E_ARG;
  NgpParams P;
  fill_model(ctx, P);
  hipLaunchKernelGGL(ngp_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, pos, dir, n,
                     out);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

// Carves the renderer's scratch for `rays` rays out of one allocation (grown on demand).  Per-pipe: the ray list (id,
// start, direction) sized for the pipe's whole slice of the rays (every ray of a slice may hit the box: `cap` = the largest
// slice) and a counter block; the per-ray result buffers indexed by ray id are shared by the pipes of the render.
static int ensure_scratch(pxt_ngp* ctx, size_t rays, int n_pipe) {
  size_t cap = (rays + (size_t)n_pipe - 1) / (size_t)n_pipe + 2 * kTile;
  if (ctx->scratch && ctx->scratch_rays >= rays && ctx->scratch_cap >= cap && ctx->scratch_pipes >= n_pipe) return PXT_OK;
  // grow only: a context that alternates between pipe counts keeps the larger layout
  cap = std::max(cap, ctx->scratch_cap);
  n_pipe = std::max(n_pipe, ctx->scratch_pipes);
  rays = std::max(rays, ctx->scratch_rays);
  if (ctx->scratch) {
    hipError_t e0 = hipDeviceSynchronize();
    (void)e0;
    (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
  }
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  struct Offs { size_t rid, t0, dir, cnt; } o[pxt_ngp::kMaxPipes];
  for (int w = 0; w < n_pipe; ++w) {
    o[w].rid = take(cap * 4); o[w].t0 = take(cap * 4); o[w].dir = take(cap * 16);
    o[w].cnt = take(kCtrWords * sizeof(int));
  }
  const size_t o_sppd = take(rays * 4), o_spp = take(rays * 16);
  hipError_t e = hipMalloc(&ctx->scratch, off);
  if (e != hipSuccess) { set_last_error("hipMalloc(ngp scratch)", e); ctx->scratch_rays = 0; return PXT_E_HIP; }
  char* b = (char*)ctx->scratch;
  for (int w = 0; w < n_pipe; ++w) {
    NgpWork& W = ctx->work[w];
    W.rid = (unsigned*)(b + o[w].rid);
    W.t0 = (float*)(b + o[w].t0);
    W.dir = (float4*)(b + o[w].dir);
    W.counters = (int*)(b + o[w].cnt);
    W.sppbuf = (float4*)(b + o_spp);
    W.sppbuf_d = (float*)(b + o_sppd);
  }
  ctx->scratch_rays = rays;
  ctx->scratch_cap = cap;
  ctx->scratch_pipes = n_pipe;
  ctx->counters_clean = false;
  return PXT_OK;
}

// The view part of a render's parameter record (everything but the device-side camera source), with the argument checks.
static int fill_view(const pxt_ngp* ctx, const pxt_ngp_view* v, int mode, float* out_rgba, float* out_depth, uint64_t* stats,
                     uint8_t* out_u8, uint8_t* out_nz, NgpParams& P, size_t& rays) {
  if (!ctx || !v) return PXT_E_ARG;
  // every image the mode produces needs somewhere to go: its float form or its 8-bit stand-in
  if (mode != 1 && !out_rgba && !out_u8) return PXT_E_ARG;
  if (mode == 1 && !out_rgba && !out_nz) return PXT_E_ARG;
  if (mode == 2 && !out_depth && !out_nz) return PXT_E_ARG;
  if (v->width < 1 || v->height < 1 || v->spp < 1 || !(v->focal > 0.f)) return PXT_E_ARG;
  fill_model(ctx, P);
  for (int i = 0; i < 12; ++i) P.cam[i] = v->cam[i];
  P.focal = v->focal;
  P.k1 = v->k1;
  for (int i = 0; i < 3; ++i) { P.lo[i] = v->aabb_min[i]; P.hi[i] = v->aabb_max[i]; }
  for (int i = 0; i < 4; ++i) P.bg[i] = v->background[i];
  P.min_T = v->min_transmittance;
  P.W = v->width; P.H = v->height; P.spp = v->spp; P.mode = mode;
  P.out = out_rgba;
  P.out_depth = (mode == 2) ? out_depth : nullptr;
  P.out_u8 = out_u8;
  P.out_nz = out_nz;
  P.stats = (unsigned long long*)stats;
  // padded to whole 4x2 pixel blocks (the enumeration order of enum_ray)
  rays = (size_t)((v->width + 3) / 4 * 4) * ((v->height + 1) / 2 * 2) * v->spp;
  if (rays > 0x7fffffffull / kK) return PXT_E_ARG;
  return PXT_OK;
}

// ---- a chain of K renders: raygen -> render -> resolve, each ONE launch for all pipes of all K renders.
namespace {

struct ChainRender {
  pxt_ngp* ctx;
  NgpParams P;   // the whole view
  size_t rays;
};

int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  return e ? std::min(std::max(atoi(e), lo), hi) : dflt;
}

struct NgpStageSlot {
  NgpBatchItem* host = nullptr;
  hipEvent_t copied = nullptr;
};
constexpr int kNgpStageSlots = 4;
constexpr int kChainItems = kMaxChainPipes + PXT_NGP_MAX_BATCH;  // pipe records, then one record per render (resolve)

void launch_render(int modes, const NgpItemsByValue<4>* pv, const NgpBatchItem* pm, dim3 grid, int rays_per_wg, hipStream_t s) {
  const dim3 blk(256);
#define PXT_LAUNCH_RENDER(M)                                                                       \
  if (pv) hipLaunchKernelGGL(ngp_render_kernel_v<M>, grid, blk, 0, s, *pv, rays_per_wg);           \
  else hipLaunchKernelGGL(ngp_render_kernel_m<M>, grid, blk, 0, s, pm, rays_per_wg);
  switch (modes) {
    case 0: PXT_LAUNCH_RENDER(0) break;
    case 1: PXT_LAUNCH_RENDER(1) break;
    case 2: PXT_LAUNCH_RENDER(2) break;
    default: PXT_LAUNCH_RENDER(3) break;
  }
#undef PXT_LAUNCH_RENDER
}

// ws_dev: device memory for the parameter records when they do not fit the kernel-argument segment (more than 4 pipes).
int run_chain(ChainRender* R, int K, hipStream_t s0, void* ws_dev) {
  // Workgroups of the render kernel per pipe (PXT_NGP_GRID) and rays per workgroup below which fewer take part
  // (PXT_NGP_GRID_DIV): measured on the benchmark view (640 x 480 x 8 spp, ~345 k rays in the list, 4 waves per SIMD = 1024
  // resident workgroups): 2048 / 3072 / 4096 / 6144 / 8192 workgroups = 0.75 / 0.71 / 0.68 / 0.71 / 0.74 ms per render
  // (profiles/r06_experiments.md).  The grid is the queue: more waves than are resident, each with a short share.
  static const int g_render = env_int("PXT_NGP_GRID", 16384, 64, 16384), g_div = env_int("PXT_NGP_GRID_DIV", 64, 1, 1 << 20),
                   g_raygen = env_int("PXT_NGP_GRID_RAYGEN", 1024, 64, 8192);
  static const int env_pipes = env_int("PXT_NGP_PIPES", 1, 1, pxt_ngp::kMaxPipes);
  struct Pipe { int render, w; };
  Pipe pipes[kMaxChainPipes];
  int n_pipes = 0, n_per[PXT_NGP_MAX_BATCH];
  for (int k = 0; k < K; ++k) {
    pxt_ngp* ctx = R[k].ctx;
    n_per[k] = std::min(std::max(ctx->pipelines > 0 ? ctx->pipelines : env_pipes, 1), pxt_ngp::kMaxPipes);
    if (n_pipes + n_per[k] > kMaxChainPipes) return PXT_E_ARG;
    if (const int rc = ensure_scratch(ctx, R[k].rays, n_per[k])) return rc;
    for (int w = 0; w < n_per[k]; ++w, ++n_pipes) pipes[n_pipes] = {k, w};
  }
  const bool by_value = n_pipes <= 4;
  if (!by_value && !ws_dev) return PXT_E_ARG;
  // the records
  static thread_local NgpStageSlot stage[16][kNgpStageSlots];
  static thread_local int stage_next[16] = {0};
  NgpItemsByValue<4> pv{}, rv{};  // (by_value: pipes / renders)
  NgpBatchItem* rec = nullptr;    // (!by_value: the pinned staging block, pipes then renders)
  NgpStageSlot* slot = nullptr;
  if (!by_value) {
    int dev_id = 0;
    PXT_HIP_CHECK(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 16) return PXT_E_ARG;
    slot = &stage[dev_id][stage_next[dev_id]];
    stage_next[dev_id] = (stage_next[dev_id] + 1) % kNgpStageSlots;
    if (!slot->host) {
      PXT_HIP_CHECK(hipHostMalloc((void**)&slot->host, kChainItems * sizeof(NgpBatchItem), hipHostMallocDefault));
      PXT_HIP_CHECK(hipEventCreateWithFlags(&slot->copied, hipEventDisableTiming));
    } else {
      PXT_HIP_CHECK(hipEventSynchronize(slot->copied));
    }
    rec = slot->host;
  }
  int modes = R[0].P.mode, max_pixels = 0;
  for (int k = 0; k < K; ++k) {
    pxt_ngp* ctx = R[k].ctx;
    if (R[k].P.mode != modes) modes = 3;
    max_pixels = std::max(max_pixels, R[k].P.W * R[k].P.H);
    NgpBatchItem& ri = by_value ? rv.it[k] : rec[n_pipes + k];
    ri.P = R[k].P;
    ri.P.enum_lo = 0;
    ri.P.enum_hi = (long long)R[k].rays;
    ri.W = ctx->work[0];
    ri.zl.n = std::min(ctx->scratch_pipes, 4);
    for (int w = 0; w < 4; ++w) ri.zl.p[w] = w < ri.zl.n ? ctx->work[w].counters : nullptr;
    if (!ctx->counters_clean)
      for (int w = 0; w < ctx->scratch_pipes; ++w)
        PXT_HIP_CHECK(hipMemsetAsync(ctx->work[w].counters, 0, kCtrWords * sizeof(int), s0));
    ctx->counters_clean = false;  // (an error return below leaves them to the next render's memsets)
  }
  for (int p = 0; p < n_pipes; ++p) {
    const int k = pipes[p].render, w = pipes[p].w, np = n_per[k];
    NgpBatchItem& it = by_value ? pv.it[p] : rec[p];
    it.P = R[k].P;
    const long long total = (long long)R[k].rays;
    const long long per = np > 1 ? ((total / np + kTile - 1) / kTile) * kTile : total;
    it.P.enum_lo = std::min(total, per * w);
    it.P.enum_hi = (w == np - 1) ? total : std::min(total, per * (w + 1));
    it.W = R[k].ctx->work[w];
    it.zl.n = 0;
  }
  const NgpBatchItem *pm = nullptr, *rm = nullptr;
  if (!by_value) {
    PXT_HIP_CHECK(hipMemcpyAsync(ws_dev, rec, (size_t)(n_pipes + K) * sizeof(NgpBatchItem), hipMemcpyHostToDevice, s0));
    PXT_HIP_CHECK(hipEventRecord(slot->copied, s0));
    pm = (const NgpBatchItem*)ws_dev;
    rm = pm + n_pipes;
  }
  const dim3 blk(256);
  if (by_value) hipLaunchKernelGGL(ngp_raygen_kernel_v, dim3(g_raygen, n_pipes), blk, 0, s0, pv);
  else hipLaunchKernelGGL(ngp_raygen_kernel_m, dim3(g_raygen, n_pipes), blk, 0, s0, pm);
  // the render launch is the one the timing events bracket (bench.py's roofline)
  pxt_ngp* tctx = R[0].ctx;
  const bool timed = K == 1 && tctx->timing > 0 && (tctx->renders++ % tctx->timing) == 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    if (tctx->pool.empty()) {
      PXT_HIP_CHECK(hipEventCreate(&e0));
      PXT_HIP_CHECK(hipEventCreate(&e1));
    } else {
      e0 = tctx->pool.back().first;
      e1 = tctx->pool.back().second;
      tctx->pool.pop_back();
    }
    PXT_HIP_CHECK(hipEventRecord(e0, s0));
  }
  launch_render(modes, by_value ? &pv : nullptr, pm, dim3(g_render, n_pipes), g_div, s0);
  if (e0) {
    PXT_HIP_CHECK(hipEventRecord(e1, s0));
    tctx->events.emplace_back(e0, e1);
  }
  const dim3 rgrid((max_pixels + 255) / 256, K);
  if (by_value) hipLaunchKernelGGL(ngp_resolve_kernel_v, rgrid, blk, 0, s0, rv);
  else hipLaunchKernelGGL(ngp_resolve_kernel_m, rgrid, blk, 0, s0, rm);
  PXT_HIP_CHECK(hipGetLastError());
  for (int k = 0; k < K; ++k) R[k].ctx->counters_clean = true;
  return PXT_OK;
}

}  // namespace

static int render_impl(pxt_ngp* ctx, const pxt_ngp_view* v, int mode, float* out_rgba, float* out_depth,
                       uint64_t* stats, void* stream, const float* pose_src = nullptr, const PoseConv* conv = nullptr,
                       float* cam_out = nullptr, uint8_t* out_u8 = nullptr, uint8_t* out_nz = nullptr,
                       bool camera_from_slot = false) {
  ChainRender R;
  R.ctx = ctx;
  if (const int rcv = fill_view(ctx, v, mode, out_rgba, out_depth, stats, out_u8, out_nz, R.P, R.rays)) return rcv;
  if (pose_src) {  // the camera is derived on the device, in stream order, from a pose record the host has not seen
    if (!conv) return PXT_E_ARG;
    hipLaunchKernelGGL(ngp_pose_to_camera_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pose_src, *conv, ctx->cam_dev,
                       cam_out);
    R.P.cam_dev = ctx->cam_dev;
  } else if (camera_from_slot) {  // written by an earlier kernel of this stream (the LM kernel's epilogue)
    R.P.cam_dev = ctx->cam_dev;
  }
  return run_chain(&R, 1, (hipStream_t)stream, nullptr);
}

extern "C" int pxt_ngp_render(pxt_ngp* ctx, const pxt_ngp_view* v, float* out_rgba, uint64_t* stats,
                              void* stream) {
  if (!v || !out_rgba || (v->mode != 0 && v->mode != 1)) return PXT_E_ARG;
  return render_impl(ctx, v, v->mode, out_rgba, nullptr, stats, stream);
}

extern "C" int pxt_ngp_render_both(pxt_ngp* ctx, const pxt_ngp_view* v, float* out_rgba, float* out_depth_rgba,
                                   uint64_t* stats, void* stream) {
  if (!out_depth_rgba || !out_rgba) return PXT_E_ARG;
  return render_impl(ctx, v, 2, out_rgba, out_depth_rgba, stats, stream);
}

// out_depth_rgba != NULL: Shade + Depth in one march (pxt_ngp_render_both); NULL: one render in view->mode.
extern "C" int pxt_ngp_render_both_from_pose(pxt_ngp* ctx, const pxt_ngp_view* v, const float* pose12,
                                             const double* conv27, float* cam_out13, float* out_rgba,
                                             float* out_depth_rgba, uint64_t* stats, void* stream) {
  if (!v || !pose12 || !conv27) return PXT_E_ARG;
  if (!out_depth_rgba && v->mode != 0 && v->mode != 1) return PXT_E_ARG;
  if (!out_rgba) return PXT_E_ARG;
  const PoseConv cv = make_pose_conv(conv27);
  return render_impl(ctx, v, out_depth_rgba ? 2 : v->mode, out_rgba, out_depth_rgba, stats, stream, pose12, &cv, cam_out13);
}

extern "C" float* pxt_ngp_camera_slot(pxt_ngp* ctx) { return ctx ? ctx->cam_dev : nullptr; }

extern "C" int pxt_ngp_render_frame(pxt_ngp* ctx, const pxt_ngp_view* v, int32_t mode, int32_t camera_from_slot,
                                    const pxt_ngp_outputs* out, uint64_t* stats, void* stream) {
  if (!v || !out || mode < 0 || mode > 2) return PXT_E_ARG;
  return render_impl(ctx, v, mode, out->rgba, out->depth_rgba, stats, stream, nullptr, nullptr, nullptr, out->rgb_u8,
                     out->depth_nz, camera_from_slot != 0);
}

extern "C" int64_t pxt_ngp_batch_workspace_bytes(int32_t n_renders) {
  if (n_renders < 1 || n_renders > PXT_NGP_MAX_BATCH) return PXT_E_ARG;
  return (int64_t)((size_t)(pxt_ngp::kMaxPipes + 1) * n_renders * sizeof(NgpBatchItem) + 255) / 256 * 256;  // pipes + 1 resolve record each
}

// K renders of K contexts - a frame's Depth + Shade pair, or K objects tracked in lock-step - as ONE chain of three launches.
extern "C" int pxt_ngp_render_frame_batch(pxt_ngp* const* ctxs, const pxt_ngp_view* views, int32_t n_renders,
                                          const int32_t* modes, int32_t camera_from_slot, const pxt_ngp_outputs* outs,
                                          uint64_t* const* stats, void* batch_workspace, void* stream) {
  if (!ctxs || !views || !outs || !modes) return PXT_E_ARG;
  if (n_renders < 1 || n_renders > PXT_NGP_MAX_BATCH) return PXT_E_ARG;
  if (n_renders > 2 && !batch_workspace) return PXT_E_ARG;
  const int K = n_renders;
  for (int a = 0; a < K; ++a) {
    if (!ctxs[a] || modes[a] < 0 || modes[a] > 2) return PXT_E_ARG;
    for (int b = a + 1; b < K; ++b)  // a context owns ONE set of ray lists
      if (ctxs[a] == ctxs[b]) return PXT_E_ARG;
  }
  ChainRender R[PXT_NGP_MAX_BATCH];
  for (int k = 0; k < K; ++k) {
    R[k].ctx = ctxs[k];
    if (const int rcv = fill_view(ctxs[k], &views[k], modes[k], outs[k].rgba, outs[k].depth_rgba, stats ? stats[k] : nullptr,
                                  outs[k].rgb_u8, outs[k].depth_nz, R[k].P, R[k].rays))
      return rcv;
    if (camera_from_slot) R[k].P.cam_dev = ctxs[k]->cam_dev;
  }
  return run_chain(R, K, (hipStream_t)stream, batch_workspace);
}

extern "C" int pxt_ngp_set_pipelines(pxt_ngp* ctx, int32_t n) {
  if (!ctx || n < 0 || n > pxt_ngp::kMaxPipes) return PXT_E_ARG;
  ctx->pipelines = n;
  return PXT_OK;
}

extern "C" int pxt_ngp_timing_enable(pxt_ngp* ctx, int32_t enable) {
  if (!ctx) return PXT_E_ARG;
  ctx->timing = enable > 0 ? enable : 0;
  ctx->renders = 0;
  // the event pairs of the first timed renders exist before any of them runs: creating a dozen events inside a render's
  // enqueue is host time on the frame's critical path (the instrument must not slow the first frame it measures)
  while (ctx->timing > 0 && ctx->pool.size() < 24) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    PXT_HIP_CHECK(hipEventCreate(&e0));
    PXT_HIP_CHECK(hipEventCreate(&e1));
    ctx->pool.emplace_back(e0, e1);
  }
  return PXT_OK;
}

extern "C" int pxt_ngp_timing_read(pxt_ngp* ctx, float* total_ms, int32_t* n_launches) {
  if (!ctx || !total_ms || !n_launches) return PXT_E_ARG;
  float tot = 0.f;
  for (auto& ev : ctx->events) {
    PXT_HIP_CHECK(hipEventSynchronize(ev.second));
    float ms = 0.f;
    PXT_HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
    tot += ms;
    ctx->pool.push_back(ev);
  }
  *total_ms = tot;
  *n_launches = (int32_t)ctx->events.size();
  ctx->events.clear();
  return PXT_OK;
}
