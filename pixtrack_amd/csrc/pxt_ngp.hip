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
//  * One wavefront owns an 8x8 pixel tile; lane = ray.  All `spp` passes of the tile
//    run in that wave, so the spp average needs no atomics and is order-deterministic.
//  * Per march step each lane finds ITS next occupied sample (bitfield DDA, divergent),
//    then the wave evaluates the 64 samples together:
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

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace pxt {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

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
  const half8* wfrag;          // [kNumFrags][64] fragment-ordered weights
  const uint8_t* occ;          // bitfield
  NgpLevel lv[kMaxLevels];
  int n_levels, cascades;
  float aabb_scale, cone_angle, depth_scale, dt_lo, dt_hi;
  // view
  float cam[12];
  float focal, k1;
  float lo[3], hi[3];
  float bg[4];
  float min_T;
  int W, H, spp, mode;
  float* out;
  unsigned long long* stats;
  int ablate;  // debug only (PXT_NGP_ABLATE): 1 skip gathers, 2 skip MLPs
};

__device__ inline float calc_dt(float t, float cone, float lo, float hi) {
  return fminf(fmaxf(t * cone, lo), hi);
}

__device__ inline int mip_from_pos(float x, float y, float z, int cascades) {
  float m = fmaxf(fabsf(x - 0.5f), fmaxf(fabsf(y - 0.5f), fabsf(z - 0.5f)));
  int e;
  frexpf(m, &e);
  return min(max(e + 1, 0), cascades - 1);
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

// ReLU + fp16 pack of 8 consecutive accumulator rows -> one B fragment.
__device__ inline half8 relu_pack8(const f32x16& a, int base, bool relu) {
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = a[base + j];
    if (relu) v = fmaxf(v, 0.f);
    r[j] = (half_t)v;
  }
  return r;
}

// Hash-grid encode + both MLPs for the wave's 64 samples (lane = sample).  Must be called
// by all 64 lanes (MFMA); `alive` only gates the gathers.  ux,uy,uz: position in [0,1]^3.
__device__ inline void ngp_eval(const NgpParams& P, const half8* s_w, int lane, bool alive, float ux,
                                float uy, float uz, const unsigned* shB0, const unsigned* shB1,
                                float& logit, float* rgbv) {
        // ---- hash grid encode (lane = sample) ----
        unsigned Flo[8], Fhi[8];
        {
#pragma unroll
          for (int l = 0; l < kMaxLevels; ++l) {
            float f0 = 0.f, f1 = 0.f;
            if (alive && l < P.n_levels && !(P.ablate & 1)) {
              const NgpLevel& Lv = P.lv[l];
              const float qx = ux * Lv.scale + 0.5f, qy = uy * Lv.scale + 0.5f, qz = uz * Lv.scale + 0.5f;
              const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
              const float ax = qx - fx, ay = qy - fy, az = qz - fz;
              const unsigned gx = (unsigned)(int)fx, gy = (unsigned)(int)fy, gz = (unsigned)(int)fz;
              unsigned vals[8];
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const unsigned cx = gx + (c & 1), cy = gy + ((c >> 1) & 1), cz = gz + ((c >> 2) & 1);
                unsigned idx;
                if (Lv.hashed)
                  idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
                else
                  idx = cx + cy * Lv.res + cz * Lv.res * Lv.res;
                idx = idx % Lv.size + Lv.offset;
                vals[c] = P.grid[idx];
              }
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
            }
            const unsigned pk = pack_h2(f0, f1);
            if (l < 8) Flo[l] = pk; else Fhi[l - 8] = pk;
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) swap32(Flo[i], Fhi[i]);
        // B fragments of the feature input: [cb][q]
        half8 xB[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          xB[0][q] = as_half8(Flo[4 * q], Flo[4 * q + 1], Flo[4 * q + 2], Flo[4 * q + 3]);
          xB[1][q] = as_half8(Fhi[4 * q], Fhi[4 * q + 1], Fhi[4 * q + 2], Fhi[4 * q + 3]);
        }

        if (P.ablate & 2) {
          logit = 10.f + __builtin_bit_cast(float, xB[0][0][0] == (half_t)3.f ? 1u : 0u);
          rgbv[0] = rgbv[1] = rgbv[2] = 0.5f;
          return;
        }
        // ---- density MLP: 32 -> 64 (ReLU) -> 16 ----
        f32x16 h1[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x16 a = {0};
#pragma unroll
            for (int q = 0; q < 2; ++q)
              a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragD1 + 2 * rb + q) * 64 + lane], xB[cb][q], a, 0, 0, 0);
            h1[rb][cb] = a;
          }
        f32x16 dout[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x16 a = {0};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragD2 + q) * 64 + lane],
                                                       relu_pack8(h1[q >> 1][cb], 8 * (q & 1), true), a, 0, 0, 0);
          dout[cb] = a;
        }
        // ---- colour MLP: [16 density outputs | 16 SH] -> 64 -> 64 -> 16 ----
        f32x16 c1[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x16 a = {0};
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC1 + 2 * rb) * 64 + lane],
                                                       relu_pack8(dout[cb], 0, false), a, 0, 0, 0);
            const half8 shb = cb == 0 ? as_half8(shB0[0], shB0[1], shB0[2], shB0[3])
                                      : as_half8(shB1[0], shB1[1], shB1[2], shB1[3]);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC1 + 2 * rb + 1) * 64 + lane], shb, a, 0, 0, 0);
            c1[rb][cb] = a;
          }
        f32x16 c2[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x16 a = {0};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC2 + 4 * rb + q) * 64 + lane],
                                                         relu_pack8(c1[q >> 1][cb], 8 * (q & 1), true), a, 0, 0, 0);
            c2[rb][cb] = a;
          }
        f32x16 cout[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x16 a = {0};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(kFragC3 + q) * 64 + lane],
                                                       relu_pack8(c2[q >> 1][cb], 8 * (q & 1), true), a, 0, 0, 0);
          cout[cb] = a;
        }
        // rows 0..3 of column block cb sit in the low half; bring block 1 to the high lanes
        unsigned r0 = __builtin_bit_cast(unsigned, dout[0][0]), r1 = __builtin_bit_cast(unsigned, dout[1][0]);
        swap32(r0, r1);
        logit = __builtin_bit_cast(float, r0);
        // The activation is applied BEFORE the cross-lane move: v_permlane32_swap reading a
        // register an MFMA is still writing returned stale rows (regs > 0) on gfx950/ROCm 7.2;
        // a VALU op in between gets the MFMA->VALU wait states the compiler does model.
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          unsigned a0 = __builtin_bit_cast(unsigned, 1.0f / (1.0f + expf(-cout[0][c])));
          unsigned a1 = __builtin_bit_cast(unsigned, 1.0f / (1.0f + expf(-cout[1][c])));
          swap32(a0, a1);
          rgbv[c] = __builtin_bit_cast(float, a0);
        }
}

__global__ __launch_bounds__(256) void ngp_render_kernel(const NgpParams P) {
  __shared__ half8 s_w[kNumFrags * 64];
  for (int i = threadIdx.x; i < kNumFrags * 64; i += 256) s_w[i] = P.wfrag[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = (P.W + 7) / 8, tiles_y = (P.H + 7) / 8;
  const int n_tiles = tiles_x * tiles_y;
  const float half_s = P.aabb_scale * 0.5f;
  const float scene_lo = 0.5f - half_s, inv_s = 1.0f / P.aabb_scale;
  unsigned long long n_samples = 0, n_batches = 0, n_hit = 0;

  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int px = (tile % tiles_x) * 8 + (lane & 7), py = (tile / tiles_x) * 8 + (lane >> 3);
    const bool inb = px < P.W && py < P.H;
    // ---- ray through the pixel centre (oracle generate_rays) ----
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
    float d[3], o[3], idir[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      d[a] = (dxn * P.cam[4 * a + 0] + dyn * P.cam[4 * a + 1]) + P.cam[4 * a + 2];
      o[a] = P.cam[4 * a + 3];
    }
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      d[a] = d[a] / nrm;
      idir[a] = 1.0f / d[a];
    }
    const float fn = sqrtf((P.cam[2] * P.cam[2] + P.cam[6] * P.cam[6]) + P.cam[10] * P.cam[10]);
    const float zdot = (d[0] * (P.cam[2] / fn) + d[1] * (P.cam[6] / fn)) + d[2] * (P.cam[10] / fn);
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float lo = fmaxf(P.lo[a], scene_lo), hi = fminf(P.hi[a], 0.5f + half_s);
      const float t0 = (lo - o[a]) * idir[a], t1 = (hi - o[a]) * idir[a];
      tmin = fmaxf(tmin, fminf(t0, t1));
      tmax = fminf(tmax, fmaxf(t0, t1));
    }
    const bool hit = inb && (tmax > fmaxf(tmin, 0.f));
    if (!__any(hit)) {
      if (inb) {
        float4 r;
        r.x = P.bg[0] * P.bg[3]; r.y = P.bg[1] * P.bg[3]; r.z = P.bg[2] * P.bg[3]; r.w = P.bg[3];
        *(float4*)(P.out + 4 * ((size_t)py * P.W + px)) = r;
      }
      continue;
    }
    // ---- SH of the view direction, moved into the two 32-column operand halves ----
    unsigned shB0[4], shB1[4];
    {
      float sh[16];
      sh4_eval(d[0], d[1], d[2], sh);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        shB0[i] = pack_h2(sh[2 * i], sh[2 * i + 1]);          // coeffs 0..7
        shB1[i] = pack_h2(sh[8 + 2 * i], sh[8 + 2 * i + 1]);  // coeffs 8..15
        swap32(shB0[i], shB1[i]);
      }
    }
    const unsigned pix_index = (unsigned)py * (unsigned)P.W + (unsigned)px;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_a = 0.f;

    for (int s = 0; s < P.spp; ++s) {
      float t = fmaxf(tmin, 0.f) + 1e-6f;
      {
        unsigned h = pix_index * 747796405u + (unsigned)s * 2891336453u + 1u;
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
        const float uj = (float)(h >> 8) * (1.0f / 16777216.0f);
        t = t + uj * calc_dt(t, P.cone_angle, P.dt_lo, P.dt_hi);
      }
      bool alive = hit;
      if (alive) ++n_hit;
      float T = 1.f, cr = 0.f, cg = 0.f, cb_ = 0.f, ca = 0.f;

      while (__any(alive)) {
        // ---- per-lane search for the next occupied sample ----
        float pos[3] = {0.5f, 0.5f, 0.5f};
        float dt = P.dt_lo;
        if (alive) {
          bool found = false;
          for (int guard = 0; guard < 100000; ++guard) {
            if (t >= tmax) break;
#pragma unroll
            for (int a = 0; a < 3; ++a) pos[a] = o[a] + t * d[a];
            dt = calc_dt(t, P.cone_angle, P.dt_lo, P.dt_hi);
            int e;
            frexpf(dt * (float)kGrid, &e);
            const int mip = min(P.cascades - 1, max(e, mip_from_pos(pos[0], pos[1], pos[2], P.cascades)));
            const float msc = ldexpf(1.0f, -mip);
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
            const bool occ = inside && ((P.occ[lin >> 3] >> (lin & 7u)) & 1u);
            if (occ) {
              found = true;
              break;
            }
            // advance_to_next_voxel: step in dt increments past the cell border
            const float res = ldexpf((float)kGrid, -mip);
            float tm = INFINITY;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float p = res * (pos[a] - 0.5f);
              const float sg = d[a] > 0.f ? 1.f : (d[a] < 0.f ? -1.f : 0.f);
              const float tx = (floorf(p + 0.5f + 0.5f * sg) - p) * idir[a];
              if (d[a] != 0.f) tm = fminf(tm, tx);
            }
            const float t_target = t + fmaxf(tm / res, 0.f);
            do {
              t = t + calc_dt(t, P.cone_angle, P.dt_lo, P.dt_hi);
            } while (t < t_target);
          }
          alive = found;
        }
        if (!__any(alive)) break;
        n_batches += 1;
        if (alive) n_samples += 1;

        float logit, rgbv[3];
        ngp_eval(P, s_w, lane, alive, (pos[0] - scene_lo) * inv_s, (pos[1] - scene_lo) * inv_s,
                 (pos[2] - scene_lo) * inv_s, shB0, shB1, logit, rgbv);
        // ---- composite ----
        if (alive) {
          const float density = expf(logit);
          if (P.mode == 1) {
            const float depth = (t * zdot) * P.depth_scale;
            rgbv[0] = rgbv[1] = rgbv[2] = depth;
          }
          const float alpha = 1.0f - expf(-density * dt);
          const float wgt = alpha * T;
          cr += wgt * rgbv[0];
          cg += wgt * rgbv[1];
          cb_ += wgt * rgbv[2];
          ca += wgt;
          T = T * (1.0f - alpha);
          if (T < P.min_T) {
            cr /= ca; cg /= ca; cb_ /= ca; ca = 1.0f;
            alive = false;
          }
          t = t + dt;
        }
      }
      acc_r += cr; acc_g += cg; acc_b += cb_; acc_a += ca;
    }
    if (inb) {
      const float inv = 1.0f / (float)P.spp;
      float4 r;
      r.w = acc_a * inv;
      r.x = acc_r * inv + P.bg[0] * P.bg[3] * (1.0f - r.w);
      r.y = acc_g * inv + P.bg[1] * P.bg[3] * (1.0f - r.w);
      r.z = acc_b * inv + P.bg[2] * P.bg[3] * (1.0f - r.w);
      r.w = r.w + P.bg[3] * (1.0f - r.w);
      *(float4*)(P.out + 4 * ((size_t)py * P.W + px)) = r;
    }
  }
  if (P.stats) {
    // wave totals -> 3 atomics per wave
    for (int m = 32; m >= 1; m >>= 1) {
      n_samples += __shfl_xor(n_samples, m, 64);
      n_hit += __shfl_xor(n_hit, m, 64);
    }
    if (lane == 0) {
      atomicAdd(P.stats + 0, n_samples);
      atomicAdd(P.stats + 1, n_hit);
      atomicAdd(P.stats + 2, n_batches);
    }
  }
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
  ngp_eval(P, s_w, lane, ok, (pos[3 * ii] - scene_lo) * inv_s, (pos[3 * ii + 1] - scene_lo) * inv_s,
           (pos[3 * ii + 2] - scene_lo) * inv_s, shB0, shB1, logit, rgbv);
  if (ok) {
    out[4 * i] = logit;
    out[4 * i + 1] = rgbv[0];
    out[4 * i + 2] = rgbv[1];
    out[4 * i + 3] = rgbv[2];
  }
}

}  // namespace pxt

struct pxt_ngp {
  pxt_ngp_model model;
  unsigned* grid = nullptr;
  pxt::half8* wfrag = nullptr;
  uint8_t* occ = nullptr;
  pxt::NgpLevel lv[pxt::kMaxLevels];
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
  hipError_t e = hipMalloc((void**)&ctx->grid, (size_t)off * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->wfrag, frag.size() * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->occ, (size_t)n_occ_bytes);
  if (e == hipSuccess) e = hipMemcpy(ctx->grid, grid_params, (size_t)off * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ctx->wfrag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ctx->occ, occupancy, (size_t)n_occ_bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_last_error("pxt_ngp_create", e);
    pxt_ngp_destroy(ctx);
    return PXT_E_HIP;
  }
  *out_ctx = ctx;
  return PXT_OK;
}

extern "C" int pxt_ngp_destroy(pxt_ngp* ctx) {
  if (!ctx) return PXT_E_ARG;
  if (ctx->grid) hipFree(ctx->grid);
  if (ctx->wfrag) hipFree(ctx->wfrag);
  if (ctx->occ) hipFree(ctx->occ);
  delete ctx;
  return PXT_OK;
}

static void fill_model(const pxt_ngp* ctx, NgpParams& P) {
  std::memset(&P, 0, sizeof(P));
  P.grid = ctx->grid;
  P.wfrag = ctx->wfrag;
  P.occ = ctx->occ;
  for (int l = 0; l < kMaxLevels; ++l) P.lv[l] = ctx->lv[l < ctx->model.n_levels ? l : 0];
  P.n_levels = ctx->model.n_levels;
  P.cascades = ctx->model.grid_cascades;
  P.aabb_scale = ctx->model.aabb_scale;
  P.cone_angle = ctx->model.cone_angle;
  P.depth_scale = ctx->model.depth_scale;
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

extern "C" int pxt_ngp_render(pxt_ngp* ctx, const pxt_ngp_view* v, float* out_rgba, uint64_t* stats,
                              void* stream) {
  if (!ctx || !v || !out_rgba) return PXT_E_ARG;
  if (v->width < 1 || v->height < 1 || v->spp < 1 || !(v->focal > 0.f)) return PXT_E_ARG;
  NgpParams P;
  fill_model(ctx, P);
  P.grid = ctx->grid;
  P.wfrag = ctx->wfrag;
  P.occ = ctx->occ;
  for (int l = 0; l < kMaxLevels; ++l) P.lv[l] = ctx->lv[l < ctx->model.n_levels ? l : 0];
  P.n_levels = ctx->model.n_levels;
  P.cascades = ctx->model.grid_cascades;
  P.aabb_scale = ctx->model.aabb_scale;
  P.cone_angle = ctx->model.cone_angle;
  P.depth_scale = ctx->model.depth_scale;
  P.dt_lo = (float)(std::sqrt(3.0) / 1024.0);
  P.dt_hi = P.dt_lo * (float)(1 << (P.cascades - 1)) * (float)(1024 / kGrid);
  for (int i = 0; i < 12; ++i) P.cam[i] = v->cam[i];
  P.focal = v->focal;
  P.k1 = v->k1;
  for (int i = 0; i < 3; ++i) { P.lo[i] = v->aabb_min[i]; P.hi[i] = v->aabb_max[i]; }
  for (int i = 0; i < 4; ++i) P.bg[i] = v->background[i];
  P.min_T = v->min_transmittance;
  P.W = v->width; P.H = v->height; P.spp = v->spp; P.mode = v->mode;
  P.out = out_rgba;
  P.stats = (unsigned long long*)stats;
  { const char* e = getenv("PXT_NGP_ABLATE"); P.ablate = e ? atoi(e) : 0; }
  const int n_tiles = ((v->width + 7) / 8) * ((v->height + 7) / 8);
  int grid = (n_tiles + 3) / 4;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(ngp_render_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}
