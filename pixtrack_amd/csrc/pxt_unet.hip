// VGG16-UNet feature pyramid (pixloc `UNet`) for gfx950.
//
// Replaces `pred = self.model({"image": image_tensor})`
// (pixtrack/localization/feature_extractor.py:48, prepare_input :31-32).
//
// Data layout: activations are NHWC fp16 in HBM (one pixel's channels contiguous), so
//  * an MFMA operand fragment (8 consecutive input channels of one pixel / of one
//    filter tap) is ONE 16-byte LDS read,
//  * the 1x1 heads and the LM kernel read whole pixels as contiguous records.
// The 3x3 convolutions are implicit GEMMs on v_mfma_f32_32x32x16_f16 with fp32
// accumulation: rows = output channels (A operand = filter taps), columns = pixels
// (B operand = the shifted input window), K = 9 * Cin walked as (Cin chunk of 32) x
// (9 taps) x (2 k-steps of 16).  The kernel is in pxt_conv_v2.h: the input halo of a chunk is
// double-buffered in LDS, the filter taps arrive pre-packed in fragment order straight from L2.
// Bias (or the folded BatchNorm affine), ReLU and the encoder's 2x2 max-pool are fused into the
// epilogue.
#include "pxt_common.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace pxt {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kNumConv = 17;
constexpr int kNumHeads = 3;

// ---------------------------------------------------------------------------
// First layer: image (HWC, 0..255, float or u8) [* mask] -> /255 -> ImageNet
// normalisation -> conv3x3 (3 -> Cout) + bias + ReLU -> fp16 NHWC.  HBM-bound.
// ---------------------------------------------------------------------------
// A workgroup owns a 64 x 4 pixel strip: the normalised 66 x 6 x 3 input window is computed once
// into LDS; then thread = pixel, all Cout channels, 16 at a time.  The filter index is
// wave-uniform, so the 27 x Cout weights arrive as scalar loads (SGPR operands of the FMAs):
// no LDS or vector-memory traffic for them at all.  The results go through an XOR-swizzled LDS
// tile so that the global stores are contiguous (a row of the strip is 64 px x Cout x 2 B in a
// row): thread = pixel stores were 16-B pieces 2*Cout bytes apart and ran at 1.1 TB/s.
// All images of a batch run in ONE launch (blockIdx.y = image; type and mask per image).
constexpr int kFW = 64, kFH = 4;
struct FirstImages {
  const void* image[PXT_UNET_MAX_BATCH];
  const uint8_t* mask[PXT_UNET_MAX_BATCH];
  int is_u8[PXT_UNET_MAX_BATCH];
};

template <int COUT>
__global__ __launch_bounds__(256) void conv_first_kernel(const FirstImages imgs, int H, int W,
                                                         const float* __restrict__ wts,
                                                         const float* __restrict__ bias,
                                                         half_t* __restrict__ out) {
  __shared__ float s_px[(kFH + 2) * (kFW + 2) * 3];
  __shared__ __attribute__((aligned(16))) half_t s_out[kFH * kFW * COUT];
  const int im = blockIdx.y;
  const void* image = imgs.image[im];
  const uint8_t* mask = imgs.mask[im];
  const bool u8 = imgs.is_u8[im] != 0;
  out += (size_t)im * H * W * COUT;
  const int tiles_x = (W + kFW - 1) / kFW;
  const int tx0 = (blockIdx.x % tiles_x) * kFW, ty0 = (blockIdx.x / tiles_x) * kFH;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
  for (int i = threadIdx.x; i < (kFH + 2) * (kFW + 2); i += 256) {
    const int hy = i / (kFW + 2), hx = i % (kFW + 2);
    const int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
    const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
    float m = 1.f;
    if (ok && mask) m = (float)mask[(size_t)yy * W + xx];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = 0.f;
      if (ok) {
        const size_t idx = ((size_t)yy * W + xx) * 3 + c;
        float raw = u8 ? (float)((const uint8_t*)image)[idx] : ((const float*)image)[idx];
        if (mask) raw *= m;
        v = (raw / 255.0f - mean[c]) * istd[c];
      }
      s_px[i * 3 + c] = v;
    }
  }
  __syncthreads();
  // One wave per strip row (64 pixels = two 32-pixel MFMA column blocks), all 64 output channels, K = 27 taps padded
  // to 32 = two k-steps of v_mfma_f32_32x32x16_f16.  The fp32 operands are split into fp16 halves, x = xh + xl and
  // w = wh + wl (xh = fp16(x), xl = fp16(x - xh): 22 of the 24 mantissa bits), and the product is formed as
  // wh xh + wl xh + wh xl with fp32 accumulation - three MFMAs at the fp16 rate instead of the exact-f32
  // v_mfma_f32_32x32x2_f32 chain of round 2, which ran at the f32 VECTOR rate (1/16) and took 3.6k matrix cycles per
  // 64 pixels: 52 us for both images.  The dropped wl xl term is < 2^-22 of a product: far below the fp16 rounding of
  // the layer's output.
  static_assert(COUT == 64, "two 32-row blocks");
  const int lane = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int r31 = lane & 31, khalf = lane >> 5;
  half8 wh[2][2], wl[2][2];  // A operand: row = output channel 32 cb + r31, k = 16 s + 8 khalf + j
  int koff[2][8];            // s_px offset (floats, relative to the pixel's window origin) of tap k; -1: padding
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int k = 16 * s2 + 8 * khalf + jj;  // (ky, kx, c) = (k / 9, (k / 3) % 3, k % 3)
      koff[s2][jj] = k < 27 ? ((k / 9) * (kFW + 2) + (k / 3) % 3) * 3 + k % 3 : -1;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const float w = k < 27 ? wts[(size_t)(32 * cb + r31) * 27 + k] : 0.f;
        const half_t h = (half_t)w;
        wh[cb][s2][jj] = h;
        wl[cb][s2][jj] = (half_t)(w - (float)h);
      }
    }
  constexpr int kPieces = COUT / 8;
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int lx = 32 * pb + r31;
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = bias[32 * cb + (r & 3) + 8 * (r >> 2) + 4 * khalf];
    const float* win = s_px + (ly * (kFW + 2) + lx) * 3;
    half8 xh[2], xl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float v = koff[s2][jj] >= 0 ? win[koff[s2][jj]] : 0.f;
        const half_t h = (half_t)v;
        xh[s2][jj] = h;
        xl[s2][jj] = (half_t)(v - (float)h);
      }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cb][s2], xh[s2], acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cb][s2], xh[s2], acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cb][s2], xl[s2], acc[cb], 0, 0, 0);
      }
    // D[row = channel][col = pixel]: this lane holds pixel lx, channels (r&3) + 8 (r>>2) + 4 khalf of each block;
    // ReLU, fp16, into the pixel's record of the XOR-swizzled LDS tile (16-B pieces of 8 channels)
    const int t = ly * kFW + lx;
    half_t* rec = s_out + (size_t)t * COUT;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (half_t)fmaxf(acc[cb][4 * g + j], 0.f);
        const int piece = 4 * cb + g;  // channels 32 cb + 8 g .. + 7; this lane owns the half 4 khalf .. + 3
        *(half4*)(rec + ((piece ^ (t & (kPieces - 1))) << 3) + 4 * khalf) = o;
      }
  }
  __syncthreads();
  // coalesced write-out: the strip's row `ry` is kFW * COUT contiguous halves in global memory
#pragma unroll
  for (int k = 0; k < kFH * kFW * kPieces / 256; ++k) {
    const int q = threadIdx.x + 256 * k;       // 16-B piece of the tile, row-major [row][px][piece]
    const int piece = q % kPieces, px = (q / kPieces) % kFW, ry = q / (kPieces * kFW);
    const int x = tx0 + px, y = ty0 + ry;
    if (x < W && y < H) {
      const int t = ry * kFW + px;  // the thread that computed this pixel
      *(half8*)(out + ((size_t)y * W + x) * COUT + piece * 8) =
          *(const half8*)(s_out + (size_t)t * COUT + ((piece ^ (t & (kPieces - 1))) << 3));
    }
  }
}

// ---------------------------------------------------------------------------
// 3x3 convolution, pad 1, NHWC fp16 -> NHWC fp16, MFMA implicit GEMM: pxt_conv_v2.h.
// ---------------------------------------------------------------------------
// Decoder input cat([bilinear x2 upsample(prev) (align_corners=False), skip[:2Hp, :2Wp]]) formed
// while staging (UPCAT): channels [0, Cp) of the conv input are interpolated from `prev`
// [Hp][Wp][Cp], the rest come from `in` = skip [Hs][Ws][Cin - Cp].  Same fp32 lerp and fp16
// rounding as a materialised concat would give.
struct UpSrc {
  const half_t* prev;
  int Hp, Wp, Cp, Hs, Ws;
};

}  // namespace pxt
#include "pxt_conv_v2.h"
#include "pxt_conv_v3.h"
namespace pxt {

// Split-K epilogue: out = relu(sum_z partial[z] + bias) -> fp16, 4 channels per thread.
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splits, long long n4, int Cout,
                                     const float* __restrict__ bias, int relu, half_t* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const size_t e = (size_t)i * 4;
  float4 a = *(const float4*)(partial + e);
  for (int z = 1; z < splits; ++z) {
    const float4 b = *(const float4*)(partial + (size_t)z * n4 * 4 + e);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  const float4 bv = *(const float4*)(bias + (e % Cout));
  a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
  if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  half4 o;
  o[0] = (half_t)a.x; o[1] = (half_t)a.y; o[2] = (half_t)a.z; o[3] = (half_t)a.w;
  *(half4*)(out + e) = o;
}

// ---------------------------------------------------------------------------
// 1x1 heads on MFMA: rows = output channels (descriptor C, then the uncertainty row),
// columns = 32 pixels per wave, K = Cin read straight from the NHWC map (one 16-B load per
// lane per k-step).  Epilogue: optional per-pixel L2 normalisation of the descriptor,
// confidence = sigmoid(-x), float32 HWC record [C | conf | 0 pad].
// wts: fp16 [32*NT][Cin] (rows >= C+1 are zero), bias fp32 [32*NT].
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void head_mfma_body(const half_t* __restrict__ in, long long npix, int Cin,
                                               const half_t* __restrict__ wts, const float* __restrict__ bias, int Cout,
                                               int normalize, float* __restrict__ out, int cstride, long long wave) {
  const int lane = threadIdx.x & 63;
  const long long p0 = wave * 32;
  if (p0 >= npix) return;
  const int r31 = lane & 31, khalf = lane >> 5;
  const long long pix = min(p0 + r31, npix - 1);
  f32x16 acc[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const half_t* xp = in + (size_t)pix * Cin + 8 * khalf;
  const half_t* wp = wts + (size_t)r31 * Cin + 8 * khalf;
  // Cin % 32 == 0 for every head (32 / 128 / 512): two k-steps per trip keep twice the loads
  // in flight (the coarsest head is pure load latency: 38 waves, K = 512).
  for (int k0 = 0; k0 < Cin; k0 += 32) {
    const half8 b0 = *(const half8*)(xp + k0), b1 = *(const half8*)(xp + k0 + 16);
    half8 a0[NT], a1[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      a0[c] = *(const half8*)(wp + (size_t)(32 * c) * Cin + k0);
      a1[c] = *(const half8*)(wp + (size_t)(32 * c) * Cin + k0 + 16);
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[c], b0, acc[c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[c], b1, acc[c], 0, 0, 0);
  }
  // bias, then the descriptor's squared norm over this lane's rows and the partner half's
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const float v = acc[c][r] + bias[co];
      acc[c][r] = v;
      if (co < Cout) ss += v * v;
    }
  ss += __shfl_xor(ss, 32, 64);
  const float inv = normalize ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 1.f;
  if (p0 + r31 >= npix) return;
  float* o = out + (size_t)(p0 + r31) * cstride;
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = 32 * c + 8 * g + 4 * khalf;
      if (co >= cstride) continue;
      float4 v;
      float* vv = (float*)&v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = acc[c][4 * g + j];
        const int cj = co + j;
        vv[j] = (cj < Cout) ? x * inv : (cj == Cout ? 1.f / (1.f + expf(x)) : 0.f);
      }
      *(float4*)(o + co) = v;
    }
}

template <int NT>
__global__ __launch_bounds__(256) void head_mfma_kernel(const half_t* __restrict__ in, long long npix, int Cin,
                                                        const half_t* __restrict__ wts,
                                                        const float* __restrict__ bias, int Cout, int normalize,
                                                        float* __restrict__ out, int cstride) {
  head_mfma_body<NT>(in, npix, Cin, wts, bias, Cout, normalize, out, cstride,
                     ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
}

// The two coarse heads of one image (strides 16 and 4: 38 + 600 waves at 640x480) in ONE launch of one-wave
// workgroups: workgroups below a.waves serve job a, the rest job b (round 4: one dispatch less per pass).
struct HeadJob {
  const half_t* in;
  long long npix;
  int Cin;
  const half_t* wts;
  const float* bias;
  int Cout, normalize;
  float* out;
  int cstride;
  long long waves;
};
template <int NT>
__global__ __launch_bounds__(64) void head_pair_kernel(const HeadJob a, const HeadJob b) {
  const long long w = blockIdx.x;
  if (w < a.waves) head_mfma_body<NT>(a.in, a.npix, a.Cin, a.wts, a.bias, a.Cout, a.normalize, a.out, a.cstride, w);
  else head_mfma_body<NT>(b.in, b.npix, b.Cin, b.wts, b.bias, b.Cout, b.normalize, b.out, b.cstride, w - a.waves);
}

// ... and of a BATCH of images (round 5: the lock-step multi-object pass): blockIdx.y = image.  One launch instead of
// 2 x n_img (a 16-image pass spent 1.5 ms of side-stream time in 32 one-image head launches squeezed between the
// decoder's workgroups, profiles/r05_experiments.md).
struct HeadBatch {
  HeadJob a, b;  // image 0's; image i reads in + i * npix * Cin
  float* out_a[PXT_UNET_MAX_BATCH];
  float* out_b[PXT_UNET_MAX_BATCH];
  int normalize[PXT_UNET_MAX_BATCH];
};
template <int NT>
__global__ __launch_bounds__(64) void head_batch_kernel(const HeadBatch hb) {
  const long long w = blockIdx.x;
  const int img = blockIdx.y;
  if (w < hb.a.waves)
    head_mfma_body<NT>(hb.a.in + (size_t)img * hb.a.npix * hb.a.Cin, hb.a.npix, hb.a.Cin, hb.a.wts, hb.a.bias, hb.a.Cout,
                       hb.normalize[img], hb.out_a[img], hb.a.cstride, w);
  else
    head_mfma_body<NT>(hb.b.in + (size_t)img * hb.b.npix * hb.b.Cin, hb.b.npix, hb.b.Cin, hb.b.wts, hb.b.bias, hb.b.Cout,
                       hb.normalize[img], hb.out_b[img], hb.b.cstride, w - hb.a.waves);
}

// Diagnostic (not on the frame's path): largest |x| and the number of non-finite values of an fp16 activation tensor.
// pixloc runs its UNet in fp32; here activations are stored as fp16 (65504 max).  With He-initialised synthetic weights
// they stay below ~50; whether a real VGG16 / MegaDepth checkpoint keeps every layer inside fp16's range can only be
// checked with that checkpoint - pxt_unet_activation_stats is the check (VERDICT r3 missing #5).
__global__ void activation_stats_kernel(const half_t* __restrict__ x, long long n8, float* __restrict__ out) {
  float m = 0.f;
  unsigned bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const half8 v = *(const half8*)(x + 8 * i);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j];
      if (!(fabsf(f) <= 65504.f)) ++bad;  // inf or nan
      else m = fmaxf(m, fabsf(f));
    }
  }
  for (int s = 32; s >= 1; s >>= 1) {
    m = fmaxf(m, __shfl_xor(m, s, 64));
    bad += __shfl_xor(bad, s, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax((unsigned*)out, __float_as_uint(m));  // non-negative floats order like their bit patterns
    if (bad) atomicAdd((unsigned*)out + 1, bad);
  }
}

// ---------------------------------------------------------------------------
// 2x2 max-pool stride 2 (floor), NHWC fp16, 8 channels per thread.
// ---------------------------------------------------------------------------
__global__ void maxpool2_kernel(const half_t* __restrict__ in, int H, int W, int C,
                                half_t* __restrict__ out, int Ho, int Wo, int n_img) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8;
  if (i >= (long long)n_img * Ho * Wo * c8) return;
  const int c = (int)(i % c8) * 8;
  long long p = i / c8;
  const int img = (int)(p / ((long long)Ho * Wo));
  p -= (long long)img * Ho * Wo;
  in += (size_t)img * H * W * C;
  out += (size_t)img * Ho * Wo * C;
  const int x = (int)(p % Wo), y = (int)(p / Wo);
  const half_t* s = in + ((size_t)(2 * y) * W + 2 * x) * C + c;
  half8 a = *(const half8*)s, b = *(const half8*)(s + C), d = *(const half8*)(s + (size_t)W * C),
        e = *(const half8*)(s + (size_t)W * C + C);
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    half_t m0 = a[j] > b[j] ? a[j] : b[j];
    half_t m1 = d[j] > e[j] ? d[j] : e[j];
    o[j] = m0 > m1 ? m0 : m1;
  }
  *(half8*)(out + ((size_t)y * Wo + x) * C + c) = o;
}

// ---------------------------------------------------------------------------
// Constant-tile skipping in encoder blocks 1-3 (round 5; pxt_conv_v2.h TileSkip).
//  * skip_block_or_kernel: per image an 8 x 8-pixel block grid of "the network's input is NOT the constant here":
//    a masked query is exactly 0 wherever its mask is 0 (the mask is multiplied into the image ahead of the first
//    convolution, pixloc_tracker_r9.py:224-225), a NeRF reference render is exactly 0 outside the object.  An image
//    with neither a mask nor a uint8 type counts as active everywhere.
//  * skip_tile_flags_kernel: flag per (layer, image, tile) = 1 unless the tile's dependency cone - its pixels' footprint
//    at full resolution grown by the layer's cumulative radius R - holds no active block and lies inside the image (no
//    zero padding reached through any layer).  Conservative on both counts: an extra 1 only costs the skip.
// ---------------------------------------------------------------------------
struct SkipImages {
  const void* image[PXT_UNET_MAX_BATCH];
  const uint8_t* mask[PXT_UNET_MAX_BATCH];
  int is_u8[PXT_UNET_MAX_BATCH];
};
// one thread per (block, row): 8 consecutive lanes hold a block's 8 rows and OR their results with three shuffles
__global__ void skip_block_or_kernel(const SkipImages im, int H, int W, int bh, int bw, uint8_t* __restrict__ grid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, img = blockIdx.y;
  const int b = t >> 3, r = t & 7;
  const bool live = b < bh * bw;
  const int by = live ? b / bw : 0, bx = live ? b % bw : 0;
  const uint8_t* mask = im.mask[img];
  const uint8_t* u8 = im.is_u8[img] ? (const uint8_t*)im.image[img] : nullptr;
  unsigned any = (mask == nullptr && u8 == nullptr) ? 1u : 0u;
  const int y = 8 * by + r, x0 = 8 * bx, n = min(8, W - x0);
  if (live && !any && y < H) {
    if (mask) {  // (a set mask bit counts whatever the pixel holds)
      const uint8_t* p = mask + (size_t)y * W + x0;
      if (n == 8 && (((uintptr_t)p) & 7) == 0) any = *(const unsigned long long*)p != 0ull;
      else for (int x = 0; x < n; ++x) any |= p[x];
    } else {
      const uint8_t* p = u8 + ((size_t)y * W + x0) * 3;
      if (n == 8 && (((uintptr_t)p) & 7) == 0) {
        const unsigned long long* q = (const unsigned long long*)p;
        any = (q[0] | q[1] | q[2]) != 0ull;
      } else {
        for (int x = 0; x < 3 * n; ++x) any |= p[x];
      }
    }
  }
  any |= __shfl_xor(any, 1, 64);
  any |= __shfl_xor(any, 2, 64);
  any |= __shfl_xor(any, 4, 64);
  if (live && r == 0) grid[(size_t)img * bh * bw + b] = any ? 1 : 0;
}

struct SkipLayerGeo {
  int th, shift, radius, h, w, offset;  // tile rows (x 16 columns), log2 stride, cone radius R in input pixels, layer size, flag offset
};
struct SkipGeo {
  SkipLayerGeo l[6];
  int n_layers;
};
// one wave per (layer, image, tile): the lanes share the block cells of the tile's cone
__global__ void skip_tile_flags_kernel(const SkipGeo geo, const uint8_t* __restrict__ grid, int H, int W, int bh, int bw,
                                       int n_img, uint8_t* __restrict__ flags) {
  const SkipLayerGeo g = geo.l[blockIdx.y];
  const int tiles_x = (g.w + 15) >> 4, tiles_y = (g.h + g.th - 1) / g.th;
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (t >= n_img * tiles_x * tiles_y) return;  // (wave-uniform)
  const int img = t / (tiles_x * tiles_y), tile = t % (tiles_x * tiles_y);
  const int ty0 = (tile / tiles_x) * g.th, tx0 = (tile % tiles_x) * 16;
  // the tile's pixels at full resolution, grown by R
  const int y0 = (ty0 << g.shift) - g.radius, y1 = (min(ty0 + g.th, g.h) << g.shift) + g.radius;  // [y0, y1)
  const int x0 = (tx0 << g.shift) - g.radius, x1 = (min(tx0 + 16, g.w) << g.shift) + g.radius;
  unsigned any = (y0 < 0 || x0 < 0 || y1 > H || x1 > W) ? 1u : 0u;  // zero padding inside the cone
  if (!any) {
    const uint8_t* gi = grid + (size_t)img * bh * bw;
    const int by0 = y0 >> 3, bx0 = x0 >> 3, nby = ((y1 - 1) >> 3) - by0 + 1, nbx = ((x1 - 1) >> 3) - bx0 + 1;
    for (int c = lane; c < nby * nbx; c += 64) any |= gi[(by0 + c / nbx) * bw + bx0 + c % nbx];
  }
  any = __ballot(any != 0) != 0ull;
  if (lane == 0) flags[g.offset + t] = any ? 1 : 0;
}

__global__ void skip_fill_kernel(half_t* __restrict__ dst, int n_pix, int C, const half_t* __restrict__ value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pix * C) dst[i] = value[i % C];
}
__global__ void skip_copy_kernel(half_t* __restrict__ dst, const half_t* __restrict__ src, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------
struct UnetLayer {
  int cin, cout;
  const void* w;
  const float* b;
};

}  // namespace pxt

struct pxt_unet {
  void* dev_head = nullptr;            // fp16 [32*NT][Cin] head weights + fp32 padded biases
  const pxt::half_t* head_w[pxt::kNumHeads];
  const float* head_b[pxt::kNumHeads];
  void* dev_blob = nullptr;
  void* dev_head0 = nullptr;           // fine head fused into the last decoder layer: frags | conf_w | bias
  void* dev_packed = nullptr;          // conv taps in MFMA A-fragment order (pxt_conv_v2.h)
  const pxt::half_t* conv_packed[pxt::kNumConv];
  int64_t n_bytes = 0;
  // Side streams.  A pass owns one SideSet: its coarse heads run on `side` beside the decoder.  A batch of two
  // images runs as TWO single-image passes, image 0 on the caller's stream, image 1 on `pass2` (fork / join
  // with events): the layers of a pass are 20-75 us launches whose tails the other pass fills (two-image
  // pass 0.91 -> 0.87 ms in scripts/exp_unet_two_streams.py).  PXT_UNET_STREAMS=1 keeps the batched launches.
  struct SideSet {
    hipStream_t side = nullptr;
    hipEvent_t ev_enc4 = nullptr, ev_dec1 = nullptr, ev_side = nullptr;
  } sides[2];
  hipStream_t pass2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // constant-tile skipping: the output vectors of conv layers 1..6 for the constant input, per plan signature
  struct SkipEntry {
    int cfg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long stamp = 0;  // 0: empty
    pxt::half_t* values = nullptr;  // device [6][256]
  } skip_cache[8];
  unsigned long long skip_clock = 0;
  bool tile_skip = true;         // pxt_unet_set_tile_skip
  void* skip_scratch = nullptr;  // device: raw zero image 48 x 48 x 3 f32 | in map 48 x 48 x 256 f16 | out map | pooled map
  bool plan_as_single = false;  // pxt_unet_set_batch_plan
  bool defer_join = false;    // pxt_unet_set_defer_join: the pair entry leaves the second pass un-joined ...
  bool join_pending = false;  // ... until pxt_unet_pair_join (or the next forward call) makes the caller's stream wait for it
  pxt::UnetLayer conv[pxt::kNumConv];
  pxt::UnetLayer head[pxt::kNumHeads];
};

using namespace pxt;

namespace {

// ---- launch plan of one 3x3 layer ------------------------------------------------------------
// Tile configurations of conv3x3_v2_kernel<CW, PBW, WC, WP>: a workgroup (4 waves, WC x WP) covers
// TH = 2*PBW*WP rows x 16 columns of pixels and BNC = 32*CW*WC output channels.
struct V2Cfg { int CW, PBW, WC, WP, KC, KS; };
constexpr int kNumCfgs = 22;
constexpr V2Cfg kV2Cfgs[kNumCfgs] = {{0, 0, 0, 0, 0},
                                     {2, 4, 2, 2, 32},   // 1: 16x16 px x 128 ch  (wave: 64 ch x 128 px)
                                     {2, 2, 1, 4, 32},   // 2: 16x16 px x  64 ch  (wave: 64 ch x  64 px)
                                     {0, 0, 0, 0, 0},    // 3: (32x16 px x 64 ch: measured slower everywhere, removed)
                                     {2, 2, 2, 2, 32},   // 4:  8x16 px x 128 ch
                                     {0, 0, 0, 0, 0},    // 5: (32x16 px x 32 ch: removed)
                                     {1, 2, 1, 4, 32},   // 6: 16x16 px x  32 ch
                                     {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0},
                                     // conv3x3_v3_kernel (pxt_conv_v3.h): pixel fragments shared by the vertical taps,
                                     // filter fragments through LDS
                                     {2, 4, 2, 2, 32},   // 11: 16x16 px x 128 ch
                                     {0, 0, 0, 0, 0},
                                     {2, 4, 1, 4, 16},   // 13: 32x16 px x  64 ch, 16-channel chunks
                                     {2, 2, 2, 2, 32},   // 14:  8x16 px x 128 ch
                                     {2, 3, 2, 2, 32},   // 15: 12x16 px x 128 ch (no fused pool: odd block count)
                                     {1, 4, 1, 4, 16},   // 16: 32x16 px x  32 ch, 16-channel chunks
                                     {2, 3, 1, 4, 16},   // 17: 24x16 px x  64 ch, 16-channel chunks (the decoder's 64-channel
                                                         //     layers: the 32-row tile + the upsampling staging spills)
                                     {2, 3, 2, 2, 32, 2},  // 18: as 15, eight waves: the K range split over two wave quartets
                                     {2, 4, 2, 2, 32, 2},  // 19: as 11, eight waves
                                     {1, 2, 1, 4, 16},     // 20: 16x16 px x 32 ch, 16-channel chunks: 33 KB of LDS, 4 workgroups per CU
                                     {2, 2, 1, 4, 16}};    // 21: 16x16 px x 64 ch, 16-channel chunks
inline bool cfg_valid(int cfg) { return cfg >= 1 && cfg < kNumCfgs && kV2Cfgs[cfg].CW != 0; }
inline bool cfg_v3(int cfg) { return cfg >= 11; }
inline int cfg_th(int cfg) { return 2 * kV2Cfgs[cfg].PBW * kV2Cfgs[cfg].WP; }
inline int cfg_bnc(int cfg) { return 32 * kV2Cfgs[cfg].CW * kV2Cfgs[cfg].WC; }

struct ConvPlan { int cfg, tiles, nb, splits; };

// Passes running side by side on the chip (2 while a batch of two runs as two single-image passes): the tile
// configuration, the split-K factor and the deep-ring choice look at the workgroups of ALL of them - a lone
// 120x160 layer has 160-300 workgroups and would take the one-wave-per-SIMD configuration, two of those do not
// share a CU.
thread_local int g_conv_peers = 1;
thread_local int g_conv_layer = 0;  // index of the pyramid layer being launched (0: a stand-alone call)
// pxt_unet_set_batch_plan(ctx, 1): a batch's layers take the tile configuration and the split-K factor a SINGLE image of
// the frame's two-stream pair pass would take (n_img = 1, two passes sharing the chip), whatever the batch size - so an
// image's maps are bit for bit those of the one-object tracker (the summation order of a layer is its tile's K walk and
// its split-K partition, nothing else).  0: planned for the batch as launched (fewer splits, the fastest).
thread_local bool g_plan_as_single = false;

// Split-K factor: only the smallest maps (conv5: 12 tiles per image pair) leave most of the 256 CUs
// without a workgroup; measured on the 60x80 layers (160 workgroups) every split loses to no split.
int choose_splits(int wgs, int n_chunks, bool upcat, int own = 1 << 30) {
  // (the decoder's 64-channel tiles run one workgroup per CU at 160 workgroups: split those too)
  // A pass whose OWN layer has at most 64 workgroups (512 -> 512 at 57x57: the 1/16 level of a 921x921 reference
  // render, 64 eight-wave workgroups walking 16 chunks each) is split in two even when a twin pass brings the count
  // to 128: with real assets the twin is the smaller query pass, which is elsewhere in its pyramid by then
  // (921x921 || 640x480 pair pass 1.325 -> 1.258 ms, profiles/r04_experiments.md #16; the benchmark's layers: unchanged)
  if (!upcat && wgs >= 128 && own <= 64 && n_chunks >= 8) return 2;
  if (wgs >= (upcat ? 256 : 128)) return 1;
  int splits = (256 + wgs - 1) / wgs;
  splits = std::min(splits, std::max(1, n_chunks / 4));
  return std::max(1, std::min(splits, 16));
}

ConvPlan plan_conv(int n_img, int H, int W, int cin, int cout, bool allow_split, int force_cfg = 0,
                   int force_splits = 0, bool upcat = false, bool wants_pool = false) {
  ConvPlan P;
  const int plan_img = g_plan_as_single ? 1 : n_img, plan_peers = g_plan_as_single ? 2 : g_conv_peers;
  auto wgs_of = [&](int cfg) {
    const int th = cfg_th(cfg);
    return plan_peers * plan_img * ((H + th - 1) / th) * ((W + 15) / 16) * (cout / cfg_bnc(cfg));
  };
  int cfg = force_cfg;
  if (!cfg_valid(cfg) || cout % cfg_bnc(cfg) != 0) {
    // measured per layer of the 640x480 pyramid (scripts/bench_conv.py --all-cfgs, profiles/r02_conv_cfgs.log):
    // the 8-row x 128-channel tile wins while it yields >= 2 workgroups per CU, the 16-row one below
    if (cout % 128 == 0) {
      cfg = wgs_of(4) >= 512 ? 4 : 1;
      // round 3 (profiles/r03_conv_cfgs.log): the third kernel's 12-row tile wins wherever it brings the layer
      // under ONE resident round of workgroups (two per CU) without starving it - the 120x160 and 60x80 layers:
      // 777 / 910 / 713 / 813 TFLOP/s against 705 / 800 / 686 / 808.  It has no fused max-pool (odd block count
      // per wave), so a layer whose output is pooled keeps the 8- / 16-row tile.
      const int w15 = wgs_of(15);
      if (!wants_pool && w15 >= 192 && w15 <= 512) cfg = 15;
      // ... and where even that leaves ONE four-wave workgroup per CU (the 60x80 layers), the eight-wave variants
      // put a second wave on every SIMD by splitting the K range inside the workgroup: 914 / 784 TFLOP/s.
      const bool even_k = (cin / 32) % 2 == 0;
      if (cfg == 15 && w15 <= 256 && even_k) cfg = 18;
      if (cfg == 1 && wgs_of(1) <= 256 && wgs_of(1) >= 128 && even_k) cfg = 19;
      // the 30x40 layers (a handful of tiles, split-K): the same eight-wave tile, 4 splits of 2 + 2 chunks per quartet pair
      if (cfg == 1 && allow_split && wgs_of(15) < 192 && (cin / 32) % 8 == 0) cfg = 18;
    }
    else if (cout % 64 == 0) cfg = 2;
    else cfg = 6;
    // Round 4 experiment (profiles/r04_experiments.md #18; PXT_CONV_POOL14=1): the pooled layers (last convolution of
    // encoder blocks 1-3) on the third kernel's 8-row x 128-channel tile.  With a device synchronisation after every pair
    // pass it is worth 4 % of the pass (0.826 against 0.860 ms, five interleaved runs) - and NOTHING in the frame loop,
    // where the host runs ahead of the device (bench.py: 680 / 680 / 682 against 682 / 682 / 683 frames/s, UNet stage 0.677
    // against 0.675 ms): what it shortens is the pass's start-up while the host is still enqueueing.  Off by default.
    static const bool pool14 = [] { const char* e = getenv("PXT_CONV_POOL14"); return e ? atoi(e) != 0 : false; }();
    if (pool14 && wants_pool && cout % 128 == 0) cfg = 14;
  }
  if (cfg_valid(cfg) && kV2Cfgs[cfg].KS == 2 && (upcat || (cin / 32) % 2 != 0)) cfg = cfg == 18 ? 15 : 11;
  P.cfg = cfg;
  const int th = cfg_th(cfg);
  P.tiles = n_img * ((H + th - 1) / th) * ((W + 15) / 16);
  P.nb = cout / cfg_bnc(cfg);
  const int plan_tiles = P.tiles / n_img * plan_img;
  P.splits = allow_split ? (force_splits > 0 ? std::min(force_splits, std::max(1, cin / 32)) : choose_splits(plan_peers * plan_tiles * P.nb, cin / 32, upcat, plan_tiles * P.nb)) : 1;
  if (kV2Cfgs[cfg].KS == 2) {  // every split must hold an even number of chunks: round the factor down to a divisor
    while (P.splits > 1 && (cin / 32) % (2 * P.splits) != 0) --P.splits;
  }
  return P;
}

// Tile configuration of a decoder (upsample + concat) layer.  Round 3: the third kernel's 16-channel-chunk tiles (33-39 KB
// of LDS: four workgroups per CU instead of three on these latency-bound layers) for the three high-resolution layers;
// the 60x80 one keeps its split-K tile.
inline int upcat_cfg(int cout, int H, int W) {
  return cout % 64 == 0 ? ((long long)H * W >= 120 * 160 ? 21 : 2) : 20;
}

// Bytes of split-K partials a layer may need.  The launch-time plan depends on whether the layer's output is pooled and on
// the number of passes sharing the chip (g_conv_peers), neither of which is known when the workspace is sized: the
// largest factor over those four cases.  launch_conv checks its plan against the region it is given.
size_t splitk_bytes(int n_img, int H, int W, int cin, int cout, bool upcat = false) {
  int splits = 1;
  const int peers_before = g_conv_peers;
  const bool single_before = g_plan_as_single;
  for (int single = 0; single < 2; ++single)
    for (int peers = 1; peers <= 2; ++peers)
      for (int pool = 0; pool < (upcat ? 1 : 2); ++pool) {
        g_conv_peers = peers;
        g_plan_as_single = single != 0;
        splits = std::max(splits, plan_conv(n_img, H, W, cin, cout, true, upcat ? upcat_cfg(cout, H, W) : 0, 0, upcat, pool != 0).splits);
      }
  g_conv_peers = peers_before;
  g_plan_as_single = single_before;
  return splits > 1 ? (size_t)splits * n_img * H * W * cout * sizeof(float) : 0;
}

struct Plan {
  int h[5], w[5];          // encoder block resolutions
  int dh[4], dw[4];        // decoder block output resolutions
  // byte offsets into the workspace
  size_t enc_tmp[5][2], enc_pool[5], enc_out[5], dec_out[4], splitk, splitk_bytes, total;
  size_t skip_grid, skip_flags;  // constant-tile skipping: the 8 x 8 block grid, then the tile flags of layers 1..6
  int skip_bh, skip_bw;
};

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// Workspace layout for a batch of n_img equally sized images: every activation buffer holds
// the images back to back ([n_img][h][w][C]).
bool make_plan(const pxt_unet* ctx, int n_img, int H, int W, Plan& P) {
  if (n_img < 1 || n_img > PXT_UNET_MAX_BATCH) return false;
  P.h[0] = H; P.w[0] = W;
  for (int b = 1; b < 5; ++b) { P.h[b] = P.h[b - 1] / 2; P.w[b] = P.w[b - 1] / 2; }
  if (P.h[4] < 1 || P.w[4] < 1) return false;
  int ph = P.h[4], pw = P.w[4];
  for (int d = 0; d < 4; ++d) { ph *= 2; pw *= 2; P.dh[d] = ph; P.dw[d] = pw; }
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
  static const int enc_c[5] = {64, 128, 256, 512, 512};
  static const int enc_in[5] = {3, 64, 128, 256, 512};
  for (int b = 0; b < 5; ++b) {
    const size_t px = (size_t)n_img * P.h[b] * P.w[b];
    P.enc_pool[b] = (b > 0) ? take(px * enc_in[b] * 2) : 0;
    P.enc_tmp[b][0] = take(px * enc_c[b] * 2);
    P.enc_tmp[b][1] = take(px * enc_c[b] * 2);
    P.enc_out[b] = take(px * enc_c[b] * 2);
  }
  for (int d = 0; d < 4; ++d) {
    const size_t px = (size_t)n_img * P.dh[d] * P.dw[d];
    P.dec_out[d] = take(px * ctx->conv[13 + d].cout * 2);
  }
  // one split-K partial buffer, sized for the hungriest layer
  size_t sk = 0;
  static const int blk_of[13] = {0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4};
  for (int i = 1; i < 13; ++i)
    sk = std::max(sk, splitk_bytes(n_img, P.h[blk_of[i]], P.w[blk_of[i]], ctx->conv[i].cin, ctx->conv[i].cout));
  for (int d = 0; d < 4; ++d)
    sk = std::max(sk, splitk_bytes(n_img, P.dh[d], P.dw[d], ctx->conv[13 + d].cin, ctx->conv[13 + d].cout, true));
  P.splitk = take(sk + 256);
  P.splitk_bytes = sk;
  P.skip_bh = (H + 7) / 8;
  P.skip_bw = (W + 7) / 8;
  P.skip_grid = take((size_t)n_img * P.skip_bh * P.skip_bw);
  // tiles of a layer: at most (h / 8 + 1) x (w / 16 + 1) (the smallest tile is 8 rows), six layers at strides 1, 2, 2, 4, 4, 4
  P.skip_flags = take((size_t)n_img * 6 * ((size_t)(H / 8 + 2) * (W / 16 + 2)));
  P.total = off;
  return true;
}

template <int CW, int PBW, int WC, int WP, bool UPCAT, int AR = 3, bool FIRST = false>
void launch_v2(const ConvArgs& a, dim3 grid, hipStream_t s) {
  constexpr int lds = 2 * (2 * PBW * WP + 2) * kV2RowBytes + (UPCAT ? (PBW * WP + 2) * 10 * 64 : 0) + 32 * CW * WC * 4 +
                      (FIRST ? (2 * PBW * WP + 4) * 20 * 3 * 4 : 0);
  static bool attr_done = false;  // the double-buffered halo of the 32-row tiles exceeds the 64 KiB default
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3x3_v2_kernel<CW, PBW, WC, WP, UPCAT, AR, FIRST>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv3x3_v2_kernel<CW, PBW, WC, WP, UPCAT, AR, FIRST>), grid, dim3(256), lds, s, a);
}

template <int CW, int PBW, int WC, int WP, int KC, bool UPCAT = false, int KS = 1>
void launch_v3(const ConvArgs& a, dim3 grid, hipStream_t s) {
  constexpr int lds = v3_lds_bytes(CW, PBW, WC, WP, KC, UPCAT, KS);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3x3_v3_kernel<CW, PBW, WC, WP, KC, UPCAT, KS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv3x3_v3_kernel<CW, PBW, WC, WP, KC, UPCAT, KS>), grid, dim3(256 * KS), lds, s, a);
}

void launch_v2_cfg(int cfg, bool upcat, const ConvArgs& a, dim3 grid, hipStream_t s) {
  if (cfg_v3(cfg) && upcat) {
    switch (cfg) {
      case 17: launch_v3<2, 3, 1, 4, 16, true>(a, grid, s); break;
      case 20: launch_v3<1, 2, 1, 4, 16, true>(a, grid, s); break;
      case 21: launch_v3<2, 2, 1, 4, 16, true>(a, grid, s); break;
      default: launch_v3<1, 4, 1, 4, 16, true>(a, grid, s); break;  // 16
    }
    return;
  }
  if (cfg_v3(cfg)) {
    switch (cfg) {
      case 11: launch_v3<2, 4, 2, 2, 32>(a, grid, s); break;
      case 13: launch_v3<2, 4, 1, 4, 16>(a, grid, s); break;
      case 14: launch_v3<2, 2, 2, 2, 32>(a, grid, s); break;
      case 15: launch_v3<2, 3, 2, 2, 32>(a, grid, s); break;
      case 17: launch_v3<2, 3, 1, 4, 16>(a, grid, s); break;
      case 18: launch_v3<2, 3, 2, 2, 32, false, 2>(a, grid, s); break;
      case 19: launch_v3<2, 4, 2, 2, 32, false, 2>(a, grid, s); break;
      case 20: launch_v3<1, 2, 1, 4, 16>(a, grid, s); break;
      case 21: launch_v3<2, 2, 1, 4, 16>(a, grid, s); break;
      default: launch_v3<1, 4, 1, 4, 16>(a, grid, s); break;
    }
    return;
  }
  if (upcat) {
    // decoder layers (bilinear x2 + concat formed in the staging): the 16x16-pixel tiles.  (Keep every
    // variant free of scratch: one build whose 18-step loop was not unrolled indexed its offset table
    // dynamically, 32 B of scratch per lane, and each launch - plus the head kernels on the side
    // stream - then took milliseconds waiting for scratch set-up.)
    if (cfg == 2) launch_v2<2, 2, 1, 4, true>(a, grid, s);
    else launch_v2<1, 2, 1, 4, true>(a, grid, s);
    return;
  }
  // <= 1 workgroup per CU: nothing hides a miss on the streamed taps but a deeper fragment ring
  const bool deep = (long long)g_conv_peers * grid.x * grid.y * grid.z <= 256;
  switch (cfg) {
    case 1:
      if (deep) launch_v2<2, 4, 2, 2, false, 9>(a, grid, s);
      else launch_v2<2, 4, 2, 2, false>(a, grid, s);
      break;
    case 2:
      if (a.first.enabled) launch_v2<2, 2, 1, 4, false, 3, true>(a, grid, s);
      else launch_v2<2, 2, 1, 4, false>(a, grid, s);
      break;

    case 4: launch_v2<2, 2, 2, 2, false>(a, grid, s); break;
    default: launch_v2<1, 2, 1, 4, false>(a, grid, s); break;
  }
}

// One 3x3 layer: `wpk` are the packed taps.  `pool_out` (optional) receives the 2x2 max-pool of the
// output when the layer runs without split-K; returns (through *pooled) whether it was written.
int launch_conv(int cin, int cout, const half_t* wpk, const float* bias, const half_t* in, int H, int W, half_t* out,
                hipStream_t s, int relu = 1, float* partial = nullptr, int n_img = 1, const UpSrc* up = nullptr,
                half_t* pool_out = nullptr, bool* pooled = nullptr, int force_cfg = 0, int force_splits = 0,
                const FusedHead* head = nullptr, const FusedFirst* first = nullptr, size_t partial_cap = ~(size_t)0,
                const TileSkip* skip = nullptr, int skip_cfg = 0) {
  if (cin % 32 != 0 || cout % 32 != 0) return PXT_E_ARG;
  if (up && (up->Cp % 32 != 0 || up->Cp >= cin || H != 2 * up->Hp || W != 2 * up->Wp || up->Hs < H || up->Ws < W))
    return PXT_E_ARG;
  // Experiment knob: PXT_CONV_PLAN="layer:cfg:splits;..." overrides the tile configuration / split-K factor of the
  // pyramid's convolution `layer` (1..16 in pxt_unet's order; 0 in a field = keep the default).
  {
    static const std::vector<std::array<int, 3>> plan = [] {
      std::vector<std::array<int, 3>> v;
      const char* e = getenv("PXT_CONV_PLAN");
      while (e && *e) {
        std::array<int, 3> t = {0, 0, 0};
        if (sscanf(e, "%d:%d:%d", &t[0], &t[1], &t[2]) >= 2) v.push_back(t);
        e = strchr(e, ';');
        if (e) ++e;
      }
      return v;
    }();
    for (const auto& t : plan)
      if (t[0] == g_conv_layer && g_conv_layer > 0) {
        if (t[1] > 0) force_cfg = t[1];
        if (t[2] > 0) force_splits = t[2];
      }
  }
  if (first) { force_cfg = 2; force_splits = 0; }
  if (up) {  // decoder layers: 2 / 13 (64 channels), 6 / 16 (32 channels)
    const bool ok = cout % 64 == 0 ? (force_cfg == 2 || force_cfg == 17 || force_cfg == 21) : (force_cfg == 6 || force_cfg == 16 || force_cfg == 20);
    if (!ok) force_cfg = upcat_cfg(cout, H, W);
  }
  const ConvPlan cp = plan_conv(n_img, H, W, cin, cout, partial != nullptr, force_cfg, force_splits, up != nullptr, pool_out != nullptr);
  ConvArgs a;
  // (the packed buffer holds the second kernel's layout, then the third's: pxt_conv3x3_packed_bytes)
  a.in = in; a.H = H; a.W = W; a.Cin = cin; a.wpk = cfg_v3(cp.cfg) ? wpk + (size_t)cout * 9 * cin : wpk; a.bias = bias; a.Cout = cout; a.relu = relu;
  a.out = out; a.partial = partial;
  a.up = up ? *up : UpSrc{nullptr, 0, 0, 0, 0, 0};
  a.pool = (cp.splits == 1 && !(cfg_v3(cp.cfg) && (kV2Cfgs[cp.cfg].PBW & 1))) ? pool_out : nullptr;
  if (kV2Cfgs[cp.cfg].KS == 2 && ((cin / 32) / cp.splits) % 2 != 0) return PXT_E_ARG;  // both quartets need equal K shares
  // the partials of this plan must fit the region the caller sized (a forced plan - PXT_CONV_PLAN - may ask for more)
  if (cp.splits > 1 && (size_t)cp.splits * n_img * H * W * cout * sizeof(float) > partial_cap) return PXT_E_ARG;
  std::memset(&a.head, 0, sizeof(a.head));
  if (head) {
    if (cout != 32 || cp.splits != 1 || cfg_bnc(cp.cfg) != 32) return PXT_E_ARG;
    a.head = *head;
    a.head.enabled = 1;
  }
  std::memset(&a.first, 0, sizeof(a.first));
  if (first) {  // the first layer computed in this layer's staging: the 64 -> 64 layer on its 16x16 x 64-channel tile only
    if (cin != 64 || cp.cfg != 2 || cp.splits != 1 || up) return PXT_E_ARG;
    a.first = *first;
    a.first.enabled = 1;
  }
  // constant-tile skipping: only with the very tile configuration the flags (tile geometry) and the constant vector (K
  // walk) were made for, one K pass, no fused head
  a.skip.flags = nullptr;
  a.skip.value = nullptr;
  if (skip && skip->flags && cp.splits == 1 && !head && !up && cp.cfg == skip_cfg) a.skip = *skip;
  if (pooled) *pooled = a.pool != nullptr;
  const dim3 grid(cp.tiles, cp.nb, cp.splits);
  static const bool debug_plan = getenv("PXT_CONV_DEBUG") != nullptr;
  if (debug_plan)
    fprintf(stderr, "conv layer %2d %4d->%4d @%dx%d x%d  cfg %2d  grid (%d, %d, %d)%s%s\n", g_conv_layer, cin, cout, W, H, n_img,
            cp.cfg, cp.tiles, cp.nb, cp.splits, up ? "  upcat" : "", a.pool ? "  +pool" : "");
  launch_v2_cfg(cp.cfg, up != nullptr, a, grid, s);
  if (cp.splits > 1) {
    const long long n4 = (long long)n_img * H * W * cout / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, partial, cp.splits,
                       n4, cout, bias, relu, out);
  }
  return PXT_OK;
}

// ---- constant-tile skipping: the layers' output vectors for the constant input -------------------------------
// For a plan signature (the tile configuration of conv layers 1..6) the chain c1 = conv1_2(conv1_1(0-pixel)), c2 =
// conv2_1(c1), ... is computed ONCE on the device by running each layer's own kernel configuration on a 48 x 48
// constant map and taking the centre pixel: the same K walk, the same bits as a constant tile of the real pass.
constexpr int kSkipS = 48, kSkipLayers = 6, kSkipMaxC = 256;
bool conv_plan_env_set() {
  static const bool set = getenv("PXT_CONV_PLAN") != nullptr;
  return set;
}
int ensure_skip_values(pxt_unet* ctx, const int cfg[kSkipLayers], hipStream_t s, const half_t** values) {
  ++ctx->skip_clock;
  pxt_unet::SkipEntry* victim = &ctx->skip_cache[0];
  for (auto& e : ctx->skip_cache) {
    if (e.stamp && std::memcmp(e.cfg, cfg, sizeof(e.cfg)) == 0) {
      e.stamp = ctx->skip_clock;
      *values = e.values;
      return PXT_OK;
    }
    if (e.stamp < victim->stamp) victim = &e;
  }
  // A miss happens once per plan signature (the first pass with a new set of tile configurations; eight signatures are kept).  The context's
  // passes may be running on two streams (the pair pass), and the entry being replaced may still be read there: drain
  // the device before the entry is rewritten and again before anyone uses it - a one-off bubble, no per-frame cost.
  PXT_HIP_CHECK(hipDeviceSynchronize());
  const size_t raw_b = (size_t)kSkipS * kSkipS * 3 * sizeof(float), map_b = (size_t)kSkipS * kSkipS * kSkipMaxC * sizeof(half_t);
  if (!ctx->skip_scratch) {
    PXT_HIP_CHECK(hipMalloc(&ctx->skip_scratch, raw_b + 2 * map_b));
    PXT_HIP_CHECK(hipMemset(ctx->skip_scratch, 0, raw_b + 2 * map_b));
  }
  if (!victim->values) PXT_HIP_CHECK(hipMalloc((void**)&victim->values, (size_t)kSkipLayers * kSkipMaxC * sizeof(half_t)));
  char* sc = (char*)ctx->skip_scratch;
  half_t* in_map = (half_t*)(sc + raw_b);
  half_t* out_map = (half_t*)(sc + raw_b + map_b);
  for (int k = 0; k < kSkipLayers; ++k) {
    const int li = k + 1;
    const int cin = ctx->conv[li].cin, cout = ctx->conv[li].cout;
    if (cout > kSkipMaxC || cin > kSkipMaxC || !cfg_valid(cfg[k]) || cout % cfg_bnc(cfg[k]) != 0) return PXT_E_ARG;
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.H = kSkipS; a.W = kSkipS; a.Cin = cin; a.Cout = cout; a.relu = 1;
    a.wpk = cfg_v3(cfg[k]) ? ctx->conv_packed[li] + (size_t)cout * 9 * cin : ctx->conv_packed[li];
    a.bias = ctx->conv[li].b;
    a.out = out_map;
    if (li == 1) {  // the first layer computed inside the second one's staging, from a raw zero image
      a.first.image[0] = sc;
      a.first.w = (const float*)ctx->conv[0].w;
      a.first.b = ctx->conv[0].b;
      a.first.enabled = 1;
    } else {
      const int n = kSkipS * kSkipS * cin;
      hipLaunchKernelGGL(skip_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in_map, kSkipS * kSkipS, cin,
                         (const half_t*)(victim->values + (size_t)(k - 1) * kSkipMaxC));
      a.in = in_map;
    }
    const int th = cfg_th(cfg[k]);
    const dim3 grid(((kSkipS + th - 1) / th) * (kSkipS / 16), cout / cfg_bnc(cfg[k]), 1);
    launch_v2_cfg(cfg[k], false, a, grid, s);
    hipLaunchKernelGGL(skip_copy_kernel, dim3((cout + 255) / 256), dim3(256), 0, s, victim->values + (size_t)k * kSkipMaxC,
                       (const half_t*)(out_map + ((size_t)(kSkipS / 2) * kSkipS + kSkipS / 2) * cout), cout);
  }
  PXT_HIP_CHECK(hipGetLastError());
  PXT_HIP_CHECK(hipStreamSynchronize(s));
  std::memcpy(victim->cfg, cfg, sizeof(victim->cfg));
  victim->stamp = ctx->skip_clock;
  *values = victim->values;
  return PXT_OK;
}

// Host-side packing of [Cout][3][3][Cin] fp16 taps into the kernel's fragment order.
void pack_conv_weights_host(const half_t* w, int cin, int cout, half_t* packed) {
  for (int co = 0; co < cout; ++co)
    for (int t = 0; t < 9; ++t) {
      const half_t* src = w + ((size_t)co * 9 + t) * cin;
      for (int ci = 0; ci < cin; ++ci) {
        packed[packed_weight_index(co, t, ci, cout)] = src[ci];
        packed[(size_t)cout * 9 * cin + packed_weight_index_v3(co, t, ci, cout)] = src[ci];
      }
    }
}

}  // namespace

extern "C" int pxt_unet_destroy(pxt_unet* ctx);

extern "C" int pxt_unet_create(const void* weights_host, int64_t n_bytes, pxt_unet** out_ctx) {
  if (!weights_host || !out_ctx || n_bytes < 64) return PXT_E_ARG;
  const char* p = (const char*)weights_host;
  if (std::memcmp(p, "PXTUNET1", 8) != 0) return PXT_E_ARG;
  int32_t n_conv, n_heads;
  std::memcpy(&n_conv, p + 8, 4);
  std::memcpy(&n_heads, p + 12, 4);
  if (n_conv != kNumConv || n_heads != kNumHeads) return PXT_E_ARG;
  const int32_t* dims = (const int32_t*)(p + 16);
  const int64_t* table = (const int64_t*)(p + 16 + 8 * (n_conv + n_heads));
  const int n_arrays = 2 * (n_conv + n_heads);
  for (int i = 0; i < n_arrays; ++i)
    if (table[2 * i] < 0 || table[2 * i] + table[2 * i + 1] > n_bytes || (table[2 * i] % 16) != 0)
      return PXT_E_ARG;
  pxt_unet* ctx = new pxt_unet();
  ctx->n_bytes = n_bytes;
  hipError_t e = hipMalloc(&ctx->dev_blob, (size_t)n_bytes);
  if (e != hipSuccess) { set_last_error("hipMalloc(unet weights)", e); delete ctx; return PXT_E_HIP; }
  e = hipMemcpy(ctx->dev_blob, weights_host, (size_t)n_bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_last_error("hipMemcpy(unet weights)", e); (void)hipFree(ctx->dev_blob); delete ctx; return PXT_E_HIP; }
  const char* d = (const char*)ctx->dev_blob;
  for (int i = 0; i < n_conv + n_heads; ++i) {
    UnetLayer& L = (i < n_conv) ? ctx->conv[i] : ctx->head[i - n_conv];
    L.cin = dims[2 * i];
    L.cout = dims[2 * i + 1];
    L.w = d + table[4 * i];
    L.b = (const float*)(d + table[4 * i + 2]);
    const int64_t wbytes = table[4 * i + 1], bbytes = table[4 * i + 3];
    int64_t want_w, want_b;
    if (i == 0) { want_w = (int64_t)L.cout * 27 * 4; want_b = L.cout * 4; }
    else if (i < n_conv) { want_w = (int64_t)L.cout * 9 * L.cin * 2; want_b = L.cout * 4; }
    else { want_w = (int64_t)L.cin * (L.cout + 1) * 4; want_b = (L.cout + 1) * 4; }
    if (wbytes != want_w || bbytes != want_b) { (void)hipFree(ctx->dev_blob); delete ctx; return PXT_E_ARG; }
  }
  // architecture checks (VGG16-UNet wiring the forward pass assumes)
  bool ok = ctx->conv[0].cin == 3 && ctx->conv[0].cout == 64;
  for (int i = 1; i < n_conv; ++i) ok = ok && (ctx->conv[i].cin % 32) == 0 && (ctx->conv[i].cout % 32) == 0;
  for (int i = 0; i < n_heads; ++i) ok = ok && (ctx->head[i].cin % 8) == 0 && ctx->head[i].cout + 1 <= 192;
  for (int i = 0; i < n_heads; ++i) ok = ok && (ctx->head[i].cin % 16) == 0 && ctx->head[i].cout + 1 <= 160;
  if (!ok) { (void)hipFree(ctx->dev_blob); delete ctx; return PXT_E_ARG; }
  // 3x3 taps in the conv kernel's A-fragment order
  {
    size_t total = 0;
    size_t offs[kNumConv];
    for (int i = 1; i < n_conv; ++i) { offs[i] = total; total += (size_t)2 * ctx->conv[i].cout * 9 * ctx->conv[i].cin; }  // both layouts
    std::vector<half_t> hp(total);
    for (int i = 1; i < n_conv; ++i)
      pack_conv_weights_host((const half_t*)(p + table[4 * i]), ctx->conv[i].cin, ctx->conv[i].cout, hp.data() + offs[i]);
    e = hipMalloc(&ctx->dev_packed, total * sizeof(half_t));
    if (e == hipSuccess) e = hipMemcpy(ctx->dev_packed, hp.data(), total * sizeof(half_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_last_error("packed conv weights", e); pxt_unet_destroy(ctx); return PXT_E_HIP; }
    ctx->conv_packed[0] = nullptr;
    for (int i = 1; i < n_conv; ++i) ctx->conv_packed[i] = (const half_t*)ctx->dev_packed + offs[i];
  }
  // heads: fp16 [32*NT][Cin] row-major weights (rows >= cout+1 zero) + fp32 padded bias
  {
    std::vector<char> hb;
    size_t offs_w[kNumHeads], offs_b[kNumHeads];
    for (int i = 0; i < n_heads; ++i) {
      const int cin = ctx->head[i].cin, co1 = ctx->head[i].cout + 1;
      const int rows = (co1 + 31) / 32 * 32;
      const float* Wsrc = (const float*)(p + table[4 * (n_conv + i)]);      // [cin][co1]
      const float* bsrc = (const float*)(p + table[4 * (n_conv + i) + 2]);  // [co1]
      offs_w[i] = hb.size();
      hb.resize(hb.size() + (size_t)rows * cin * sizeof(half_t));
      half_t* wd = (half_t*)(hb.data() + offs_w[i]);
      for (int r = 0; r < rows; ++r)
        for (int k = 0; k < cin; ++k) wd[(size_t)r * cin + k] = (half_t)(r < co1 ? Wsrc[(size_t)k * co1 + r] : 0.f);
      hb.resize((hb.size() + 15) / 16 * 16);
      offs_b[i] = hb.size();
      hb.resize(hb.size() + (size_t)rows * sizeof(float));
      float* bd = (float*)(hb.data() + offs_b[i]);
      for (int r = 0; r < rows; ++r) bd[r] = r < co1 ? bsrc[r] : 0.f;
      hb.resize((hb.size() + 15) / 16 * 16);
    }
    e = hipMalloc(&ctx->dev_head, hb.size());
    if (e == hipSuccess) e = hipMemcpy(ctx->dev_head, hb.data(), hb.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_last_error("head weights", e); pxt_unet_destroy(ctx); return PXT_E_HIP; }
    for (int i = 0; i < n_heads; ++i) {
      ctx->head_w[i] = (const half_t*)((const char*)ctx->dev_head + offs_w[i]);
      ctx->head_b[i] = (const float*)((const char*)ctx->dev_head + offs_b[i]);
    }
  }
  // fine head in the form the last decoder layer's epilogue consumes (pxt_conv_v2.h FusedHead)
  if (ctx->head[0].cin == 32 && ctx->head[0].cout == 32 && ctx->conv[n_conv - 1].cout == 32) {
    const float* Wsrc = (const float*)(p + table[4 * n_conv]);      // [cin 32][33]
    const float* bsrc = (const float*)(p + table[4 * n_conv + 2]);  // [33]
    std::vector<char> hb(2 * 64 * 8 * sizeof(half_t) + 32 * sizeof(float) + 36 * sizeof(float));
    half_t* fr = (half_t*)hb.data();
    float* cw = (float*)(hb.data() + 2 * 64 * 8 * sizeof(half_t));
    float* bb = cw + 32;
    for (int s2 = 0; s2 < 2; ++s2)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * s2 + j, kh = lane >> 5, row = lane & 31;
          const int ch = (r & 3) + 8 * (r >> 2) + 4 * kh;  // channel held in accumulator register r
          fr[((size_t)s2 * 64 + lane) * 8 + j] = (half_t)Wsrc[(size_t)ch * 33 + row];
        }
    for (int c = 0; c < 32; ++c) cw[c] = (float)(half_t)Wsrc[(size_t)c * 33 + 32];
    for (int i = 0; i < 33; ++i) bb[i] = bsrc[i];
    e = hipMalloc(&ctx->dev_head0, hb.size());
    if (e == hipSuccess) e = hipMemcpy(ctx->dev_head0, hb.data(), hb.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_last_error("fused head", e); pxt_unet_destroy(ctx); return PXT_E_HIP; }
  }
  *out_ctx = ctx;
  return PXT_OK;
}

extern "C" int pxt_unet_destroy(pxt_unet* ctx) {
  if (!ctx) return PXT_E_ARG;
  if (ctx->dev_blob) (void)hipFree(ctx->dev_blob);
  if (ctx->dev_head) (void)hipFree(ctx->dev_head);
  if (ctx->dev_packed) (void)hipFree(ctx->dev_packed);
  if (ctx->dev_head0) (void)hipFree(ctx->dev_head0);
  if (ctx->skip_scratch) (void)hipFree(ctx->skip_scratch);
  for (auto& e : ctx->skip_cache)
    if (e.values) (void)hipFree(e.values);
  for (auto& ss : ctx->sides) {
    if (ss.ev_enc4) (void)hipEventDestroy(ss.ev_enc4);
    if (ss.ev_dec1) (void)hipEventDestroy(ss.ev_dec1);
    if (ss.ev_side) (void)hipEventDestroy(ss.ev_side);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  delete ctx;
  return PXT_OK;
}

extern "C" int64_t pxt_unet_workspace_bytes_batch(const pxt_unet* ctx, int32_t n_images, int32_t H, int32_t W) {
  if (!ctx) return PXT_E_ARG;
  Plan P, P1;
  if (!make_plan(ctx, n_images, H, W, P) || !make_plan(ctx, 1, H, W, P1)) return 0;
  // (a batch of two may run as two single-image passes in the two halves of the workspace)
  return (int64_t)std::max(P.total, (size_t)n_images * ((P1.total + 255) / 256 * 256));
}

extern "C" int64_t pxt_unet_workspace_bytes(const pxt_unet* ctx, int32_t H, int32_t W) {
  return pxt_unet_workspace_bytes_batch(ctx, 1, H, W);
}

static int forward_pass(pxt_unet* ctx, int32_t n_images, const void* const* images,
                        const int32_t* image_is_u8, const uint8_t* const* masks, int32_t H,
                        int32_t W, float* const* out_maps, const int32_t out_cstride[3],
                        const int32_t* normalize, void* workspace, void* stream, pxt_unet::SideSet& ss) {
  if (!ctx || !images || !image_is_u8 || !out_maps || !out_cstride || !normalize || !workspace) return PXT_E_ARG;
  Plan P;
  const int B = n_images;
  if (!make_plan(ctx, B, H, W, P)) return PXT_E_ARG;
  for (int i = 0; i < B; ++i) {
    if (!images[i]) return PXT_E_ARG;
    for (int k = 0; k < 3; ++k)
      if (!out_maps[3 * i + k]) return PXT_E_ARG;
  }
  for (int k = 0; k < 3; ++k)
    if (out_cstride[k] < ctx->head[k].cout + 1 || (out_cstride[k] % 4) != 0) return PXT_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto buf = [&](size_t off) { return (half_t*)(ws + off); };
  static const int block_first[5] = {0, 2, 4, 7, 10};
  static const int block_n[5] = {2, 2, 3, 3, 3};
  if (!ss.side) {
    ss.side = pxt::shared_side_stream(&ss == &ctx->sides[0] ? 1 : 2);  // shared by all contexts (pxt_core.hip)
    if (!ss.side) return PXT_E_HIP;
    PXT_HIP_CHECK(hipEventCreateWithFlags(&ss.ev_enc4, hipEventDisableTiming));
    PXT_HIP_CHECK(hipEventCreateWithFlags(&ss.ev_dec1, hipEventDisableTiming));
    PXT_HIP_CHECK(hipEventCreateWithFlags(&ss.ev_side, hipEventDisableTiming));
  }
  // 1x1 heads at output scales 0, 2, 4 (per image: separate output tensors and normalisation
  // flags).  The two coarse ones are small, latency-bound launches whose inputs (enc4, dec1) exist
  // long before the decoder finishes: they run on a side stream beside the decoder's convolutions.
  const half_t* pre[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // fine -> coarse: dec3, dec2, dec1, dec0, enc4
  auto launch_head = [&](int k, hipStream_t hs) {
    static const int head_src[3] = {0, 2, 4};
    const UnetLayer& Lh = ctx->head[k];
    const int i = head_src[k];
    const int hh = (i == 4) ? P.h[4] : P.dh[3 - i], ww = (i == 4) ? P.w[4] : P.dw[3 - i];
    const long long npix = (long long)hh * ww;
    const long long waves = (npix + 31) / 32;
    // small maps: one wave per workgroup so the few waves spread over as many CUs
    const int tpb = waves < 1024 ? 64 : 256;
    const unsigned blocks = (unsigned)((waves * 64 + tpb - 1) / tpb);
    for (int im = 0; im < B; ++im) {
      const half_t* src = pre[i] + (size_t)im * npix * Lh.cin;
      if (Lh.cout + 1 <= 64)
        hipLaunchKernelGGL(head_mfma_kernel<2>, dim3(blocks), dim3(tpb), 0, hs, src, npix, Lh.cin, ctx->head_w[k],
                           ctx->head_b[k], Lh.cout, normalize[im], out_maps[3 * im + k], out_cstride[k]);
      else
        hipLaunchKernelGGL(head_mfma_kernel<5>, dim3(blocks), dim3(tpb), 0, hs, src, npix, Lh.cin, ctx->head_w[k],
                           ctx->head_b[k], Lh.cout, normalize[im], out_maps[3 * im + k], out_cstride[k]);
    }
  };

  // both coarse heads of a single image in one launch (after dec1 exists); PXT_UNET_MERGE_HEADS=0 keeps the two launches
  static const bool merge_heads_env = [] { const char* e = getenv("PXT_UNET_MERGE_HEADS"); return e ? atoi(e) != 0 : true; }();
  const bool merge_heads = merge_heads_env && ctx->head[1].cout + 1 > 64 && ctx->head[2].cout + 1 > 64 &&
                           ctx->head[1].cout + 1 <= 160 && ctx->head[2].cout + 1 <= 160;
  auto head_job = [&](int k) {
    static const int head_src[3] = {0, 2, 4};
    const UnetLayer& Lh = ctx->head[k];
    const int i = head_src[k];
    const int hh = (i == 4) ? P.h[4] : P.dh[3 - i], ww = (i == 4) ? P.w[4] : P.dw[3 - i];
    HeadJob j;
    j.in = pre[i]; j.npix = (long long)hh * ww; j.Cin = Lh.cin; j.wts = ctx->head_w[k]; j.bias = ctx->head_b[k];
    j.Cout = Lh.cout; j.normalize = normalize[0]; j.out = out_maps[k]; j.cstride = out_cstride[k];
    j.waves = (j.npix + 31) / 32;
    return j;
  };

  // ---- constant-tile skipping in encoder blocks 1-3 (conv layers 1..6): PXT_UNET_SKIP=0 computes every tile -------
  // Exact: a skipped tile receives the very bits it would have computed (TileSkip, pxt_conv_v2.h), so the maps do not
  // change (tests/test_unet_gpu.py compares skip on / off bit for bit).  Worth it because the tracker's images are mostly
  // constant: the masked query is 0 outside the dilated silhouette, the reference render 0 outside the object (15 % /
  // 12 % of the benchmark's 640 x 480 frame is not: 71-75 % of the first block's tiles, 66 % of the second's and 30-40 %
  // of the third's are skipped).
  TileSkip tile_skip[kSkipLayers];
  int tile_skip_cfg[kSkipLayers];
  for (int k = 0; k < kSkipLayers; ++k) { tile_skip[k].flags = nullptr; tile_skip[k].value = nullptr; tile_skip_cfg[k] = 0; }
  {
    static const bool skip_env = [] { const char* e = getenv("PXT_UNET_SKIP"); return e ? atoi(e) != 0 : true; }();
    static const bool fuse_first_env0 = [] { const char* e = getenv("PXT_UNET_FUSE_FIRST"); return e ? atoi(e) != 0 : true; }();
    bool any_source = false;
    for (int i = 0; i < B; ++i) any_source = any_source || (masks && masks[i]) || image_is_u8[i];
    const bool can = skip_env && ctx->tile_skip && !conv_plan_env_set() && fuse_first_env0 && any_source && H >= 64 && W >= 64 &&
                     ctx->conv[0].cout == 64 && ctx->conv[1].cin == 64 && ctx->conv[1].cout == 64;
    if (can) {
      static const int blk_of6[kSkipLayers] = {0, 1, 1, 2, 2, 2}, radius6[kSkipLayers] = {2, 4, 6, 10, 14, 18};
      static const bool last6[kSkipLayers] = {true, false, true, false, false, true};  // the block's last conv (fused pool)
      int cfgs[kSkipLayers];
      SkipGeo geo;
      geo.n_layers = kSkipLayers;
      int off = 0, max_tiles = 0;
      bool ok = true;
      for (int k = 0; k < kSkipLayers; ++k) {
        const int li = k + 1, b = blk_of6[k];
        // the plan launch_conv will take for this layer (same arguments: the fused first layer forces configuration 2)
        const ConvPlan cp = plan_conv(B, P.h[b], P.w[b], ctx->conv[li].cin, ctx->conv[li].cout, li != 1, li == 1 ? 2 : 0, 0, false,
                                      last6[k]);
        cfgs[k] = cp.cfg;
        ok = ok && cfg_valid(cp.cfg) && ctx->conv[li].cout <= kSkipMaxC && ctx->conv[li].cin <= kSkipMaxC;
        SkipLayerGeo& g = geo.l[k];
        g.th = cfg_th(cp.cfg); g.shift = b; g.radius = radius6[k]; g.h = P.h[b]; g.w = P.w[b]; g.offset = off;
        const int tiles = B * ((P.h[b] + g.th - 1) / g.th) * ((P.w[b] + 15) / 16);
        tile_skip_cfg[k] = cp.splits == 1 ? cp.cfg : 0;  // (a split-K layer computes every tile)
        off += tiles;
        max_tiles = std::max(max_tiles, tiles);
      }
      const half_t* values = nullptr;
      if (ok && (size_t)off <= (size_t)B * 6 * ((size_t)(H / 8 + 2) * (W / 16 + 2)) &&
          ensure_skip_values(ctx, cfgs, s, &values) == PXT_OK) {
        SkipImages si;
        for (int i = 0; i < B; ++i) { si.image[i] = images[i]; si.mask[i] = masks ? masks[i] : nullptr; si.is_u8[i] = image_is_u8[i]; }
        uint8_t* grid_b = (uint8_t*)(ws + P.skip_grid);
        uint8_t* flags_b = (uint8_t*)(ws + P.skip_flags);
        const int nb = P.skip_bh * P.skip_bw;
        hipLaunchKernelGGL(skip_block_or_kernel, dim3((8 * nb + 255) / 256, B), dim3(256), 0, s, si, H, W, P.skip_bh, P.skip_bw, grid_b);
        hipLaunchKernelGGL(skip_tile_flags_kernel, dim3((max_tiles + 3) / 4, kSkipLayers), dim3(256), 0, s, geo,
                           (const uint8_t*)grid_b, H, W, P.skip_bh, P.skip_bw, B, flags_b);
        for (int k = 0; k < kSkipLayers; ++k) {
          tile_skip[k].flags = flags_b + geo.l[k].offset;
          tile_skip[k].value = values + (size_t)k * kSkipMaxC;
        }
      }
    }
  }

  const half_t* skip[5];
  const half_t* cur = nullptr;
  bool pooled_by_conv = false;
  bool fuse_first = false;
  FusedFirst ff;
  for (int b = 0; b < 5; ++b) {
    const int h = P.h[b], w = P.w[b];
    const half_t* x;
    if (b == 0) {
      // the images differ in type (u8 render / float frame) and mask: ONE launch, blockIdx.y = image
      const UnetLayer& L0 = ctx->conv[0];
      const unsigned nblk = (unsigned)(((w + kFW - 1) / kFW) * ((h + kFH - 1) / kFH));
      half_t* o = buf(P.enc_tmp[0][0]);
      FirstImages fi;
      for (int i = 0; i < B; ++i) {
        fi.image[i] = images[i];
        fi.mask[i] = masks ? masks[i] : nullptr;
        fi.is_u8[i] = image_is_u8[i];
      }
      if (L0.cout != 64) return PXT_E_ARG;
      // The first layer is computed inside the second layer's staging (pxt_conv_v2.h FusedFirst): its 39-MB output per
      // image is never written.  PXT_UNET_FUSE_FIRST=0 keeps the separate launch (A/B, tests/test_variants_gpu.py).
      static const bool fuse_first_env = [] { const char* e = getenv("PXT_UNET_FUSE_FIRST"); return e ? atoi(e) != 0 : true; }();
      fuse_first = fuse_first_env && ctx->conv[1].cin == 64 && ctx->conv[1].cout == 64;
      if (fuse_first) {
        std::memset(&ff, 0, sizeof(ff));
        for (int i = 0; i < B; ++i) { ff.image[i] = fi.image[i]; ff.mask[i] = fi.mask[i]; ff.is_u8[i] = fi.is_u8[i]; }
        ff.w = (const float*)L0.w;
        ff.b = L0.b;
        x = nullptr;
      } else {
        hipLaunchKernelGGL(conv_first_kernel<64>, dim3(nblk, B), dim3(256), 0, s, fi, h, w, (const float*)L0.w, L0.b, o);
        x = o;
      }
    } else {
      half_t* o = buf(P.enc_pool[b]);
      if (!pooled_by_conv) {
        const int cin = ctx->conv[block_first[b]].cin;
        const long long n = (long long)B * h * w * (cin / 8);
        hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cur, P.h[b - 1],
                           P.w[b - 1], cin, o, h, w, B);
      }
      x = o;
    }
    pooled_by_conv = false;
    for (int i = (b == 0 ? 1 : 0); i < block_n[b]; ++i) {
      const bool last = i == block_n[b] - 1;
      const int li = block_first[b] + i;
      half_t* o = last ? buf(P.enc_out[b]) : buf(P.enc_tmp[b][i & 1]);
      // the block's last conv also writes the next block's pooled input (epilogue fusion)
      half_t* pool_to = (last && b < 4) ? buf(P.enc_pool[b + 1]) : nullptr;
      g_conv_layer = li;
      const bool with_first = li == 1 && fuse_first;
      const bool can_skip = li >= 1 && li <= kSkipLayers && (li != 1 || with_first);
      int rc = launch_conv(ctx->conv[li].cin, ctx->conv[li].cout, ctx->conv_packed[li], ctx->conv[li].b, x, h, w, o, s,
                           1, with_first ? nullptr : (float*)(ws + P.splitk), B, nullptr, pool_to,
                           pool_to ? &pooled_by_conv : nullptr, with_first ? 2 : 0, 0, nullptr, with_first ? &ff : nullptr,
                           P.splitk_bytes, can_skip ? &tile_skip[li - 1] : nullptr, can_skip ? tile_skip_cfg[li - 1] : 0);
      if (rc != PXT_OK) return rc;
      x = o;
    }
    skip[b] = x;
    cur = x;
  }
  // decoder
  pre[4] = skip[4];
  if (!merge_heads) {
    PXT_HIP_CHECK(hipEventRecord(ss.ev_enc4, s));
    PXT_HIP_CHECK(hipStreamWaitEvent(ss.side, ss.ev_enc4, 0));
    launch_head(2, ss.side);
  }
  const half_t* prev = skip[4];
  bool head0_fused = false;
  int ph = P.h[4], pw = P.w[4], pc = ctx->conv[12].cout;
  for (int d = 0; d < 4; ++d) {
    const UnetLayer& L = ctx->conv[13 + d];
    const int sb = 3 - d;
    const int cs = L.cin - pc;
    if (cs <= 0) return PXT_E_ARG;
    const UpSrc up{prev, ph, pw, pc, P.h[sb], P.w[sb]};  // upsample + concat happen in the conv's staging
    half_t* o = buf(P.dec_out[d]);
    FusedHead fh;
    const bool fuse_head = d == 3 && ctx->dev_head0 != nullptr && out_cstride[0] >= 36;
    if (fuse_head) {  // the fine head runs in this layer's epilogue; its own output is not materialised
      const char* hb = (const char*)ctx->dev_head0;
      fh.wfrag = (const half8*)hb;
      fh.conf_w = (const float*)(hb + 2 * 64 * 8 * sizeof(half_t));
      fh.bias = fh.conf_w + 32;
      for (int im = 0; im < B; ++im) { fh.out[im] = out_maps[3 * im]; fh.normalize[im] = normalize[im]; }
      fh.cstride = out_cstride[0];
      fh.enabled = 1;
      head0_fused = true;
    }
    g_conv_layer = 13 + d;
    int rc = launch_conv(L.cin, L.cout, ctx->conv_packed[13 + d], L.b, skip[sb], P.dh[d], P.dw[d], o, s, 1,
                         fuse_head ? nullptr : (float*)(ws + P.splitk), B, &up, nullptr, nullptr, 0, 0,
                         fuse_head ? &fh : nullptr, nullptr, P.splitk_bytes);
    if (rc != PXT_OK) return rc;
    prev = o;
    ph = P.dh[d]; pw = P.dw[d]; pc = L.cout;
    pre[3 - d] = o;
    if (d == 1) {  // dec1 feeds the stride-4 head
      PXT_HIP_CHECK(hipEventRecord(ss.ev_dec1, s));
      PXT_HIP_CHECK(hipStreamWaitEvent(ss.side, ss.ev_dec1, 0));
      if (merge_heads && B > 1) {
        HeadBatch hb;
        hb.a = head_job(2);
        hb.b = head_job(1);
        for (int im = 0; im < B; ++im) {
          hb.out_a[im] = out_maps[3 * im + 2];
          hb.out_b[im] = out_maps[3 * im + 1];
          hb.normalize[im] = normalize[im];
        }
        hipLaunchKernelGGL(head_batch_kernel<5>, dim3((unsigned)(hb.a.waves + hb.b.waves), B), dim3(64), 0, ss.side, hb);
      } else if (merge_heads) {
        const HeadJob ja = head_job(2), jb = head_job(1);
        hipLaunchKernelGGL(head_pair_kernel<5>, dim3((unsigned)(ja.waves + jb.waves)), dim3(64), 0, ss.side, ja, jb);
      } else {
        launch_head(1, ss.side);
      }
    }
  }
  g_conv_layer = 0;
  // (the heads were launched above: the two coarse ones on the side stream as soon as their input
  // existed, the fine one here)
  if (!head0_fused) launch_head(0, s);
  PXT_HIP_CHECK(hipEventRecord(ss.ev_side, ss.side));
  PXT_HIP_CHECK(hipStreamWaitEvent(s, ss.ev_side, 0));
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_unet_forward_batch(pxt_unet* ctx, int32_t n_images, const void* const* images,
                                      const int32_t* image_is_u8, const uint8_t* const* masks, int32_t H,
                                      int32_t W, float* const* out_maps, const int32_t out_cstride[3],
                                      const int32_t* normalize, void* workspace, void* stream) {
  if (!ctx || !images || !image_is_u8 || !out_maps || !out_cstride || !normalize || !workspace) return PXT_E_ARG;
  static const int n_streams = [] { const char* e = getenv("PXT_UNET_STREAMS"); return e ? atoi(e) : 2; }();
  if (ctx->join_pending) { PXT_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ctx->ev_join, 0)); ctx->join_pending = false; }
  if (n_images != 2 || n_streams < 2) {
    struct PlanMode { PlanMode(bool on) { g_plan_as_single = on; } ~PlanMode() { g_plan_as_single = false; } } plan_guard(ctx->plan_as_single && n_images > 2);
    return forward_pass(ctx, n_images, images, image_is_u8, masks, H, W, out_maps, out_cstride, normalize, workspace,
                        stream, ctx->sides[0]);
  }
  const int32_t Hs[2] = {H, H}, Ws[2] = {W, W};
  return pxt_unet_forward_pair(ctx, images, image_is_u8, masks, Hs, Ws, out_maps, out_cstride, normalize, workspace, stream);
}

extern "C" int64_t pxt_unet_workspace_bytes_pair(const pxt_unet* ctx, const int32_t H[2], const int32_t W[2]) {
  if (!ctx || !H || !W) return PXT_E_ARG;
  Plan P0, P1;
  if (!make_plan(ctx, 1, H[0], W[0], P0) || !make_plan(ctx, 1, H[1], W[1], P1)) return 0;
  return (int64_t)((P0.total + 255) / 256 * 256 + P1.total);
}

// Two images of (possibly) DIFFERENT sizes as two single-image passes side by side: the first on the caller's stream, the
// second on a side stream, in two parts of the workspace.  (The frame's reference render and its masked query: equal sizes
// in the benchmark, different ones with real assets - reference camera x 0.5 / x 0.3 - where the two passes used to run
// one after the other: 0.54 + 0.54 ms against 0.72 ms side by side.)
extern "C" int pxt_unet_forward_pair(pxt_unet* ctx, const void* const* images, const int32_t* image_is_u8,
                                     const uint8_t* const* masks, const int32_t H[2], const int32_t W[2],
                                     float* const* out_maps, const int32_t out_cstride[3], const int32_t* normalize,
                                     void* workspace, void* stream) {
  if (!ctx || !images || !image_is_u8 || !H || !W || !out_maps || !out_cstride || !normalize || !workspace) return PXT_E_ARG;
  Plan P0, P1;
  if (!make_plan(ctx, 1, H[0], W[0], P0) || !make_plan(ctx, 1, H[1], W[1], P1)) return PXT_E_ARG;
  if (!ctx->pass2) {
    ctx->pass2 = pxt::shared_side_stream(0);
    if (!ctx->pass2) return PXT_E_HIP;
    PXT_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    PXT_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  hipStream_t s = (hipStream_t)stream;
  if (ctx->join_pending) { PXT_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_join, 0)); ctx->join_pending = false; }
  PXT_HIP_CHECK(hipEventRecord(ctx->ev_fork, s));  // the second pass starts after the caller's earlier work
  PXT_HIP_CHECK(hipStreamWaitEvent(ctx->pass2, ctx->ev_fork, 0));
  const uint8_t* const no_mask[1] = {nullptr};
  struct Peers { Peers() { g_conv_peers = 2; } ~Peers() { g_conv_peers = 1; } } peers_guard;
  const size_t second = (P0.total + 255) / 256 * 256;
  for (int im = 0; im < 2; ++im) {
    int rc = forward_pass(ctx, 1, images + im, image_is_u8 + im, masks ? masks + im : no_mask, H[im], W[im], out_maps + 3 * im,
                          out_cstride, normalize + im, (char*)workspace + (im ? second : 0),
                          im == 0 ? (void*)s : (void*)ctx->pass2, ctx->sides[im]);
    if (rc != PXT_OK) return rc;
  }
  PXT_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->pass2));
  if (ctx->defer_join) {  // image 0's maps are complete in the caller's stream order; image 1's only after pxt_unet_pair_join
    ctx->join_pending = true;
    return PXT_OK;
  }
  PXT_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_join, 0));
  return PXT_OK;
}

extern "C" int pxt_unet_set_tile_skip(pxt_unet* ctx, int32_t on) {
  if (!ctx) return PXT_E_ARG;
  ctx->tile_skip = on != 0;
  return PXT_OK;
}

extern "C" int pxt_unet_set_batch_plan(pxt_unet* ctx, int32_t per_image_plan) {
  if (!ctx) return PXT_E_ARG;
  ctx->plan_as_single = per_image_plan != 0;
  return PXT_OK;
}

extern "C" int pxt_unet_set_defer_join(pxt_unet* ctx, int32_t on) {
  if (!ctx) return PXT_E_ARG;
  ctx->defer_join = on != 0;
  return PXT_OK;
}

extern "C" int pxt_unet_pair_join(pxt_unet* ctx, void* stream) {
  if (!ctx) return PXT_E_ARG;
  if (ctx->join_pending) {
    PXT_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ctx->ev_join, 0));
    ctx->join_pending = false;
  }
  return PXT_OK;
}

extern "C" int pxt_unet_forward(pxt_unet* ctx, const void* image, int32_t image_is_u8,
                                const uint8_t* mask, int32_t H, int32_t W, float* const out_maps[3],
                                const int32_t out_cstride[3], int32_t normalize, void* workspace,
                                void* stream) {
  if (!out_maps) return PXT_E_ARG;
  const void* images[1] = {image};
  const uint8_t* masks[1] = {mask};
  return pxt_unet_forward_batch(ctx, 1, images, &image_is_u8, masks, H, W, out_maps, out_cstride, &normalize,
                                workspace, stream);
}

// Per-layer range of the fp16 activations a single-image pass left in `workspace` (see activation_stats_kernel).
extern "C" int pxt_unet_activation_stats(pxt_unet* ctx, int32_t H, int32_t W, const void* workspace, float* stats,
                                         void* stream) {
  if (!ctx || !workspace || !stats) return PXT_E_ARG;
  Plan P;
  if (!make_plan(ctx, 1, H, W, P)) return PXT_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  // a pass left running by a deferred-join pair call may still be writing a workspace: wait for it first
  if (ctx->join_pending) { PXT_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_join, 0)); ctx->join_pending = false; }
  PXT_HIP_CHECK(hipMemsetAsync(stats, 0, kNumConv * 2 * sizeof(float), s));
  static const int block_first[5] = {0, 2, 4, 7, 10};
  static const int block_n[5] = {2, 2, 3, 3, 3};
  const char* ws = (const char*)workspace;
  static const bool fuse_first = [] { const char* e = getenv("PXT_UNET_FUSE_FIRST"); return e ? atoi(e) != 0 : true; }();
  std::vector<float> host_flags(kNumConv * 2, 0.f);
  for (int li = 0; li < kNumConv; ++li) {
    const half_t* x = nullptr;
    long long n = 0;
    if (li < 13) {
      int b = 4;
      while (block_first[b] > li) --b;
      const int i = li - block_first[b];
      const bool last = i == block_n[b] - 1;
      x = (const half_t*)(ws + (last ? P.enc_out[b] : P.enc_tmp[b][i & 1]));
      n = (long long)P.h[b] * P.w[b] * ctx->conv[li].cout;
      if (li == 0 && fuse_first && ctx->conv[1].cin == 64 && ctx->conv[1].cout == 64) x = nullptr;  // never materialised
    } else {
      const int d = li - 13;
      x = (const half_t*)(ws + P.dec_out[d]);
      n = (long long)P.dh[d] * P.dw[d] * ctx->conv[li].cout;
      if (d == 3 && ctx->dev_head0 != nullptr) x = nullptr;  // consumed by the fused fine head in registers
    }
    if (!x) {  // not in memory in this configuration: reported as -1 (0xBF800000: byte-wise memsets, no host source buffer
               // whose lifetime an asynchronous copy would depend on)
      PXT_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(stats + 2 * li), 0xBF800000u, 1, s));
      continue;
    }
    hipLaunchKernelGGL(activation_stats_kernel, dim3(512), dim3(256), 0, s, x, n / 8, stats + 2 * li);
  }
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int64_t pxt_conv3x3_packed_bytes(int32_t Cin, int32_t Cout) {
  if (Cin < 32 || Cout < 32 || Cin % 32 != 0 || Cout % 32 != 0) return PXT_E_ARG;
  return (int64_t)2 * Cout * 9 * Cin * (int64_t)sizeof(half_t);  // the second kernel's layout, then the third's
}

extern "C" int pxt_conv3x3_pack_weights(const void* weights, int32_t Cin, int32_t Cout, void* packed, void* stream) {
  if (!weights || !packed || pxt_conv3x3_packed_bytes(Cin, Cout) <= 0) return PXT_E_ARG;
  const long long n = (long long)Cout * 9 * Cin;
  hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)weights, Cin, Cout, (half_t*)packed);
  hipLaunchKernelGGL(pack_conv_weights_v3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)weights, Cin, Cout, (half_t*)packed + n);
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_conv3x3_packed(const void* in, int32_t H, int32_t W, int32_t Cin, const void* packed,
                                  const float* bias, int32_t Cout, int32_t relu, void* out, void* pool_out,
                                  int32_t cfg, int32_t splits, void* splitk_ws, int64_t splitk_ws_bytes,
                                  void* stream) {
  if (!in || !packed || !bias || !out || H < 1 || W < 1) return PXT_E_ARG;
  if (splits > 1 && (!splitk_ws || splitk_ws_bytes < (int64_t)splits * H * W * Cout * (int64_t)sizeof(float)))
    return PXT_E_ARG;
  if (cfg != 0 && (!cfg_valid(cfg) || Cout % cfg_bnc(cfg) != 0)) return PXT_E_ARG;
  int rc = launch_conv(Cin, Cout, (const half_t*)packed, bias, (const half_t*)in, H, W, (half_t*)out,
                       (hipStream_t)stream, relu, splits > 1 ? (float*)splitk_ws : nullptr, 1, nullptr,
                       (half_t*)pool_out, nullptr, cfg, splits > 1 ? splits : 0);
  if (rc != PXT_OK) return rc;
  PXT_HIP_CHECK(hipGetLastError());
  return PXT_OK;
}

extern "C" int pxt_conv3x3_nhwc_f16(const void* in, int32_t H, int32_t W, int32_t Cin,
                                    const void* weights, const float* bias, int32_t Cout,
                                    int32_t relu, void* out, void* stream) {
  if (!in || !weights || !bias || !out || H < 1 || W < 1) return PXT_E_ARG;
  const int64_t bytes = pxt_conv3x3_packed_bytes(Cin, Cout);
  if (bytes <= 0) return PXT_E_ARG;
  // test / profiling entry: the taps are repacked on every call into a scratch that lives as long
  // as the process (the pyramid packs once, at pxt_unet_create)
  static void* scratch = nullptr;
  static int64_t scratch_bytes = 0;
  if (scratch_bytes < bytes) {
    PXT_HIP_CHECK(hipDeviceSynchronize());
    if (scratch) (void)hipFree(scratch);
    scratch = nullptr;
    scratch_bytes = 0;
    PXT_HIP_CHECK(hipMalloc(&scratch, (size_t)bytes));
    scratch_bytes = bytes;
  }
  int rc = pxt_conv3x3_pack_weights(weights, Cin, Cout, scratch, stream);
  if (rc != PXT_OK) return rc;
  return pxt_conv3x3_packed(in, H, W, Cin, scratch, bias, Cout, relu, out, nullptr, 0, 1, nullptr, 0, stream);
}

#if PXT_EXP_STAMPS
extern "C" int pxt_debug_read_stamps(void* host, int64_t bytes) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pxt::pxt_stamps), (size_t)bytes) == hipSuccess ? 0 : -1;
}
#endif
