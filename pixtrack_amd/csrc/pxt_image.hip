// Small image kernels on the tracking path (cv2 / numpy calls in the reference).
// All are HBM-bound elementwise / stencil passes over at most a few MB.
#include "pxt_common.h"

#include <cstdlib>

namespace pxt {

// get_mask head (pixloc_tracker_r9.py:210-212 with run_vis_on_poses.py:53-54):
// depth image = float RGBA * 255 cast to uint8 (C truncation, wraps mod 256), then
// `!= 0` per channel.  All three colour channels carry the same depth, so one plane
// is kept.
__global__ void depth_nonzero_kernel(const float* __restrict__ rgba, int n, uint8_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = rgba[4 * (size_t)i] * 255.0f;
  // numpy float32 -> uint8 astype: truncate toward zero then keep the low 8 bits
  // (values are non-negative here).
  long long t = (long long)v;
  out[i] = ((t & 255) != 0) ? 1 : 0;
}

// 5x5 erosion / dilation with an all-ones kernel, OpenCV default border handling:
// erode pads with +inf (border pixels ignore the outside), dilate pads with -inf.
template <bool ERODE>
__global__ void morph5_kernel(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) return;
  uint8_t r = ERODE ? 255 : 0;
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      uint8_t v = src[(size_t)yy * W + xx];
      r = ERODE ? min(r, v) : max(r, v);
    }
  }
  dst[(size_t)y * W + x] = r;
}

// The whole mask in one pass.  n erosions (dilations) with the 5x5 box and OpenCV's default
// borders are one erosion (dilation) with the (4n+1)^2 box over the in-image pixels, and a box is
// separable, so a 64x16 tile needs the `!= 0` plane over a (re + rd) halo and four 1-D passes in
// LDS: erode rows, erode columns, dilate rows, dilate columns.  Values are 0/1: min = AND, max = OR.
constexpr int kMW = 64, kMH = 16;
__global__ __launch_bounds__(256) void depth_mask_fused_kernel(const float* __restrict__ rgba, int H, int W, int re,
                                                               int rd, uint8_t* __restrict__ out) {
  extern __shared__ uint8_t sm[];
  const int R = re + rd;
  const int aw = kMW + 2 * R, ah = kMH + 2 * R;     // input plane
  const int bw = kMW + 2 * rd;                       // after the row erosion (height ah)
  const int ch = kMH + 2 * rd;                       // after the column erosion (width bw)
  uint8_t* A = sm;                 // [ah][aw]
  uint8_t* B = A + ah * aw;        // [ah][bw]
  uint8_t* Cc = B + ah * bw;       // [ch][bw]
  uint8_t* D = Cc + ch * bw;       // [ch][kMW]
  const int x0 = blockIdx.x * kMW, y0 = blockIdx.y * kMH;
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  for (int i = tid; i < ah * aw; i += 256) {
    const int yy = y0 - R + i / aw, xx = x0 - R + i % aw;
    uint8_t v = 1;  // outside the image: neutral for the erosion
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const float f = rgba[4 * ((size_t)yy * W + xx)] * 255.0f;
      v = (((long long)f & 255) != 0) ? 1 : 0;  // numpy float32 -> uint8 astype, then != 0
    }
    A[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < ah * bw; i += 256) {
    const int r = i / bw, c = i % bw;
    uint8_t v = 1;
    for (int d = 0; d <= 2 * re; ++d) v &= A[r * aw + c + d];
    B[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < ch * bw; i += 256) {
    const int r = i / bw, c = i % bw;
    uint8_t v = 1;
    for (int d = 0; d <= 2 * re; ++d) v &= B[(r + d) * bw + c];
    const int yy = y0 - rd + r, xx = x0 - rd + c;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) v = 0;  // outside the image: neutral for the dilation
    Cc[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < ch * kMW; i += 256) {
    const int r = i / kMW, c = i % kMW;
    uint8_t v = 0;
    for (int d = 0; d <= 2 * rd; ++d) v |= Cc[r * bw + c + d];
    D[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < kMH * kMW; i += 256) {
    const int r = i / kMW, c = i % kMW;
    uint8_t v = 0;
    for (int d = 0; d <= 2 * rd; ++d) v |= D[(r + d) * kMW + c];
    const int yy = y0 + r, xx = x0 + c;
    if (yy < H && xx < W) out[(size_t)yy * W + xx] = v;
  }
}

// The same mask on BIT PLANES.  A tile row is one 128-bit word (bit j = column x0 - 32 + j: the 64 tile columns
// sit at bits 32..95, up to 32 halo columns on either side), built with two wave ballots per row; a box erosion /
// dilation along x is a few shift-AND / shift-OR steps on that word (radius 10 = shifts 1, 2, 4, 3), along y an
// AND / OR over neighbouring rows' words.  Bit for bit the result of the byte version above, which spent ~100 k
// single-byte LDS reads per tile on it (26 us per 640x480 mask, on the frame's chain between the render and the
// UNet; this one is bound by the latency of its input loads).
struct Row128 {
  unsigned long long lo, hi;
};
__device__ inline Row128 r_and(Row128 a, Row128 b) { return {a.lo & b.lo, a.hi & b.hi}; }
__device__ inline Row128 r_or(Row128 a, Row128 b) { return {a.lo | b.lo, a.hi | b.hi}; }
__device__ inline Row128 r_shl(Row128 a, int s) {  // towards higher columns; 0 < s < 64
  return {a.lo << s, (a.hi << s) | (a.lo >> (64 - s))};
}
__device__ inline Row128 r_shr(Row128 a, int s) {
  return {(a.lo >> s) | (a.hi << (64 - s)), a.hi >> s};
}

// PLANE: the input is the `!= 0` byte plane itself (written by the renderer's resolve kernel: pxt_ngp_render_frame's
// depth_nz), not the float depth image - 1 byte instead of 16 per pixel on the chain between the render and the UNet.
template <bool PLANE>
__global__ __launch_bounds__(256) void depth_mask_bits_kernel(const void* __restrict__ src, int H, int W, int re,
                                                              int rd, uint8_t* __restrict__ out) {
  const float* rgba = (const float*)src;
  const uint8_t* plane = (const uint8_t*)src;
  __shared__ Row128 sA[kMH + 2 * 16], sB[kMH + 2 * 16];
  const int R = re + rd;                 // <= 16
  const int ah = kMH + 2 * R;            // rows y0 - R .. y0 + kMH + R - 1
  const int x0 = blockIdx.x * kMW, y0 = blockIdx.y * kMH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- the `!= 0` plane: one row per wave trip, columns x0 - R .. x0 + 63 + R in two ballots; outside the
  // image: 1 (neutral for the erosion)
  for (int r = wave; r < ah; r += 4) {
    const int yy = y0 - R + r;
    unsigned long long w[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const int c = part * 64 + lane;      // column index within [0, 64 + 2R)
      const int xx = x0 - R + c;
      bool bit = true;
      if (c < kMW + 2 * R && yy >= 0 && yy < H && xx >= 0 && xx < W) {
        if (PLANE) {
          bit = plane[(size_t)yy * W + xx] != 0;
        } else {
          const float f = rgba[4 * ((size_t)yy * W + xx)] * 255.0f;
          bit = (((long long)f & 255) != 0);  // numpy float32 -> uint8 astype, then != 0
        }
      }
      w[part] = __ballot(bit);
    }
    if (lane == 0) {
      // column c sits at bit 32 - R + c
      Row128 v = {w[0], w[1]};
      sA[r] = r_shl(v, 32 - R);            // 16 <= 32 - R <= 32 (R >= 0): bits below 32 - R are zero-filled ...
      if (32 - R > 0) sA[r].lo |= (1ull << (32 - R)) - 1ull;  // ... make them neutral too
    }
  }
  __syncthreads();
  // ---- erosion along x, then along y (radius re); then clear everything outside the image (neutral for the
  // dilation) - one thread per row
  if (tid < ah) {
    Row128 a = sA[tid], e = a;
    for (int d = 1; d <= re; ++d) e = r_and(e, r_and(r_shl(a, d), r_shr(a, d)));
    // r_shl / r_shr shift zeros in at the ends of the 128-bit word: those columns are > 16 away from the tile
    sB[tid] = e;
  }
  __syncthreads();
  if (tid < ah) {
    Row128 e = {~0ull, ~0ull};
    for (int d = -re; d <= re; ++d) {
      const int q = tid + d;
      if (q >= 0 && q < ah) e = r_and(e, sB[q]);  // rows beyond the staged band are > R away from the tile
    }
    const int yy = y0 - R + tid;
    Row128 in_img = {0ull, 0ull};
    if (yy >= 0 && yy < H) {
      // columns 0 .. W-1 <-> bits 32 - x0 .. 32 - x0 + W - 1, clipped to [0, 128)
      const int b_lo = max(32 - x0, 0), b_hi = min(32 - x0 + W, 128);  // [b_lo, b_hi)
      for (int b = 0; b < 2; ++b) {
        const int lo = max(b_lo - 64 * b, 0), hi = min(b_hi - 64 * b, 64);
        unsigned long long mbits = 0ull;
        if (hi > lo) mbits = ((hi - lo) == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
        (b == 0 ? in_img.lo : in_img.hi) = mbits;
      }
    }
    sA[tid] = r_and(e, in_img);
  }
  __syncthreads();
  // ---- dilation along x (radius rd in doubling shifts), then along y
  if (tid < ah) {
    Row128 d = sA[tid];
    for (int covered = 0, step = 1; covered < rd; step *= 2) {
      const int sft = min(step, rd - covered);
      d = r_or(d, r_or(r_shl(d, sft), r_shr(d, sft)));
      covered += sft;
    }
    sB[tid] = d;
  }
  __syncthreads();
  if (tid < kMH) {
    Row128 d = {0ull, 0ull};
    const int rc = tid + R;  // this output row in the staged band
    for (int q = rc - rd; q <= rc + rd; ++q) d = r_or(d, sB[q]);  // 0 <= rc - rd, rc + rd < ah
    sA[tid] = d;
  }
  __syncthreads();
  // ---- bits 32..95 of the 16 output rows -> bytes, 4 per thread
  {
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    const Row128 d = sA[r];
    const int yy = y0 + r, xx = x0 + c4;
    if (yy < H && xx < W) {
      unsigned bytes = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = 32 + c4 + j;
        const unsigned bit = (unsigned)(((b < 64 ? d.lo >> b : d.hi >> (b - 64))) & 1ull);
        bytes |= bit << (8 * j);
      }
      if (xx + 3 < W && (W & 3) == 0) {
        *(unsigned*)(out + (size_t)yy * W + xx) = bytes;
      } else {
        for (int j = 0; j < 4 && xx + j < W; ++j) out[(size_t)yy * W + xx + j] = (uint8_t)((bytes >> (8 * j)) & 1u);
      }
    }
  }
}

// get_nerf_image tail (run_vis_on_poses.py:52-54): zero where alpha < thresh, *255,
// astype(uint8).
__global__ void rgba_to_u8_kernel(const float* __restrict__ rgba, int n, float alpha_thresh,
                                  uint8_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = *(const float4*)(rgba + 4 * (size_t)i);
  if (p.w < alpha_thresh) p.x = p.y = p.z = 0.f;
  out[3 * (size_t)i + 0] = (uint8_t)((long long)(p.x * 255.0f) & 255);
  out[3 * (size_t)i + 1] = (uint8_t)((long long)(p.y * 255.0f) & 255);
  out[3 * (size_t)i + 2] = (uint8_t)((long long)(p.z * 255.0f) & 255);
}

// cv2.resize(..., interpolation=INTER_LINEAR) on float32 HWC: half-pixel centres,
// source coordinate clamped so both taps stay inside, no antialiasing.
__global__ void resize_linear_kernel(const float* __restrict__ src, int H, int W, int C,
                                     float* __restrict__ dst, int Ho, int Wo) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= Wo || y >= Ho) return;
  const float sx = (float)W / (float)Wo, sy = (float)H / (float)Ho;
  float fx = ((float)x + 0.5f) * sx - 0.5f;
  float fy = ((float)y + 0.5f) * sy - 0.5f;
  int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
  float ax = fx - (float)x0, ay = fy - (float)y0;
  if (x0 < 0) { x0 = 0; ax = 0.f; }
  if (x0 >= W - 1) { x0 = W - 1; ax = 0.f; }
  if (y0 < 0) { y0 = 0; ay = 0.f; }
  if (y0 >= H - 1) { y0 = H - 1; ay = 0.f; }
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  for (int c = 0; c < C; ++c) {
    float a = src[((size_t)y0 * W + x0) * C + c], b = src[((size_t)y0 * W + x1) * C + c];
    float d = src[((size_t)y1 * W + x0) * C + c], e = src[((size_t)y1 * W + x1) * C + c];
    float top = a * (1.f - ax) + b * ax;
    float bot = d * (1.f - ax) + e * ax;
    dst[((size_t)y * Wo + x) * C + c] = top * (1.f - ay) + bot * ay;
  }
}

// Which pixels of a resized image can be non-zero: dst pixel (y, x) of resize_linear_kernel reads the 2 x 2 source taps
// around ((y + 0.5) H / Ho - 0.5, ...); it is exactly 0 when those taps are 0 (0 * w + 0 * w).  `active` = 1 unless every
// source pixel in the taps' neighbourhood grown by one pixel (so that no rounding of the tap position matters) is
// inactive; a source pixel is inactive where `mask` is 0 (the image was multiplied by it) or, without a mask, where the
// uint8 image is 0 in all three channels.  Feeds the UNet's constant-tile skipping for images the extractor resizes.
__global__ void resize_activity_kernel(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ u8, int H, int W, int Ho,
                                       int Wo, uint8_t* __restrict__ active) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= Wo || y >= Ho) return;
  const float sx = (float)W / (float)Wo, sy = (float)H / (float)Ho;
  const int x0 = (int)floorf(((float)x + 0.5f) * sx - 0.5f), y0 = (int)floorf(((float)y + 0.5f) * sy - 0.5f);
  unsigned any = 0;
  for (int yy = max(y0 - 1, 0); yy <= min(y0 + 2, H - 1); ++yy)
    for (int xx = max(x0 - 1, 0); xx <= min(x0 + 2, W - 1); ++xx) {
      const size_t i = (size_t)yy * W + xx;
      any |= mask ? mask[i] : (unsigned)(u8[3 * i] | u8[3 * i + 1] | u8[3 * i + 2]);
    }
  active[(size_t)y * Wo + x] = any ? 1 : 0;
}

}  // namespace pxt

using namespace pxt;

extern "C" int pxt_resize_activity(const uint8_t* mask, const uint8_t* image_u8, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                   uint8_t* active_out, void* stream) {
  if ((!mask && !image_u8) || !active_out || H < 1 || W < 1 || Ho < 1 || Wo < 1) return PXT_E_ARG;
  dim3 blk(64, 4), grd((Wo + 63) / 64, (Ho + 3) / 4);
  hipLaunchKernelGGL(resize_activity_kernel, grd, blk, 0, (hipStream_t)stream, mask, image_u8, H, W, Ho, Wo, active_out);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_depth_mask(const float* depth_rgba, int32_t H, int32_t W, int32_t n_erode,
                              int32_t n_dilate, uint8_t* mask_out, uint8_t* tmp, void* stream) {
  if (!depth_rgba || !mask_out || !tmp || H < 1 || W < 1 || n_erode < 0 || n_dilate < 0)
    return PXT_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int n = H * W;
  const int re = 2 * n_erode, rd = 2 * n_dilate, R = re + rd;
  if (R <= 16) {
    const int aw = kMW + 2 * R, ah = kMH + 2 * R, bw = kMW + 2 * rd, ch = kMH + 2 * rd;
    const size_t lds = (size_t)ah * aw + (size_t)ah * bw + (size_t)ch * bw + (size_t)ch * kMW;
    static const bool bytes_version = [] { const char* e = getenv("PXT_MASK_BYTES"); return e && atoi(e) != 0; }();
    if (bytes_version)
      hipLaunchKernelGGL(depth_mask_fused_kernel, dim3((W + kMW - 1) / kMW, (H + kMH - 1) / kMH), dim3(64, 4), lds, s,
                         depth_rgba, H, W, re, rd, mask_out);
    else
      hipLaunchKernelGGL(depth_mask_bits_kernel<false>, dim3((W + kMW - 1) / kMW, (H + kMH - 1) / kMH), dim3(256), 0, s,
                         (const void*)depth_rgba, H, W, re, rd, mask_out);
    PXT_HIP_CHECK(hipGetLastError());
    return PXT_OK;
  }
  uint8_t* a = tmp;
  uint8_t* b = tmp + n;
  hipLaunchKernelGGL(depth_nonzero_kernel, dim3((n + 255) / 256), dim3(256), 0, s, depth_rgba, n, a);
  dim3 blk(64, 4), grd((W + 63) / 64, (H + 3) / 4);
  for (int i = 0; i < n_erode; ++i) {
    hipLaunchKernelGGL(morph5_kernel<true>, grd, blk, 0, s, a, H, W, b);
    uint8_t* t = a; a = b; b = t;
  }
  for (int i = 0; i < n_dilate; ++i) {
    hipLaunchKernelGGL(morph5_kernel<false>, grd, blk, 0, s, a, H, W, b);
    uint8_t* t = a; a = b; b = t;
  }
  PXT_HIP_CHECK(hipMemcpyAsync(mask_out, a, n, hipMemcpyDeviceToDevice, s));
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_depth_mask_plane(const uint8_t* depth_nz, int32_t H, int32_t W, int32_t n_erode, int32_t n_dilate,
                                    uint8_t* mask_out, uint8_t* tmp, void* stream) {
  if (!depth_nz || !mask_out || H < 1 || W < 1 || n_erode < 0 || n_dilate < 0) return PXT_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int n = H * W;
  const int re = 2 * n_erode, rd = 2 * n_dilate, R = re + rd;
  if (R <= 16) {
    hipLaunchKernelGGL(depth_mask_bits_kernel<true>, dim3((W + kMW - 1) / kMW, (H + kMH - 1) / kMH), dim3(256), 0, s,
                       (const void*)depth_nz, H, W, re, rd, mask_out);
    PXT_HIP_CHECK(hipGetLastError());
    return PXT_OK;
  }
  if (!tmp) return PXT_E_ARG;  // larger structuring elements: the 5x5 passes one by one (2 * H * W bytes of scratch)
  const uint8_t* a = depth_nz;
  uint8_t* bufs[2] = {tmp, tmp + n};
  int k = 0;
  dim3 blk(64, 4), grd((W + 63) / 64, (H + 3) / 4);
  for (int i = 0; i < n_erode + n_dilate; ++i) {
    if (i < n_erode) hipLaunchKernelGGL(morph5_kernel<true>, grd, blk, 0, s, a, H, W, bufs[k]);
    else hipLaunchKernelGGL(morph5_kernel<false>, grd, blk, 0, s, a, H, W, bufs[k]);
    a = bufs[k];
    k ^= 1;
  }
  PXT_HIP_CHECK(hipMemcpyAsync(mask_out, a, n, hipMemcpyDeviceToDevice, s));
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_rgba_to_u8(const float* rgba, int32_t H, int32_t W, float alpha_thresh,
                              uint8_t* rgb_out, void* stream) {
  if (!rgba || !rgb_out || H < 1 || W < 1) return PXT_E_ARG;
  const int n = H * W;
  hipLaunchKernelGGL(rgba_to_u8_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     rgba, n, alpha_thresh, rgb_out);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_resize_linear(const float* src, int32_t H, int32_t W, int32_t C, float* dst,
                                 int32_t Ho, int32_t Wo, void* stream) {
  if (!src || !dst || H < 1 || W < 1 || C < 1 || Ho < 1 || Wo < 1) return PXT_E_ARG;
  dim3 blk(64, 4), grd((Wo + 63) / 64, (Ho + 3) / 4);
  hipLaunchKernelGGL(resize_linear_kernel, grd, blk, 0, (hipStream_t)stream, src, H, W, C, dst, Ho, Wo);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}
