// 3x3 convolution (pad 1), NHWC fp16 -> NHWC fp16, MFMA implicit GEMM -- second generation.
// Included by pxt_unet.hip (needs half_t / half8 / half4 / f32x16 / UpSrc from there).
//
// What changed against the first kernel (measured there: the matrix pipe waited on the
// global -> VGPR -> LDS staging of halo AND filter taps, two barriers per 32-channel chunk):
//  * the filter taps never touch LDS.  They are pre-packed once (pxt_unet_create /
//    pxt_conv3x3_pack_weights) in MFMA A-fragment order [chunk][tap][k-step][cout block][lane][8],
//    so a wave fetches the fragment of one (tap, k-step) as ONE fully coalesced 1-KiB load straight
//    into the operand registers, one step ahead of its use (they are L2 hits: every workgroup of a
//    layer reads the same few hundred KiB);
//  * LDS holds only the input halo, double buffered: the next chunk's halo is fetched at the start
//    of a chunk and written into the other buffer in the middle of it -> ONE barrier per chunk;
//  * halo pixels are 64-B records with the 16-B segment index XOR-swizzled by (column >> 2) & 3 on
//    a row pitch that is a multiple of 256 B: each 16-lane group of a ds_read_b128 fragment read
//    (16 consecutive columns) hits 16 distinct slots -> conflict free without padding;
//  * a wave owns CW x PBW MFMA tiles (up to 64 output channels x 128 pixels): one A fragment feeds
//    PBW MFMAs and one B fragment CW of them, which halves LDS and L1 traffic per MFMA;
//  * the epilogue pairs half-waves with v_permlane32_swap so every lane stores 16 contiguous bytes,
//    and can write the 2x2 max-pooled copy of its tile (the encoder's next input) as well.  (Plain
//    stores: non-temporal and write-through (sc1) output stores were measured against them to shorten
//    the end-of-kernel L2 write-back - both made the two-image pass 7 % slower, 1.04 vs 0.97 ms.)
#pragma once

#include <type_traits>

namespace pxt {

constexpr int kV2RowBytes = 1280;  // LDS pitch of a halo row: 20 pixel records (18 used) = 5 x 256 B
constexpr int kV2Cols = 18;        // tile width 16 + 2

// Optional fused epilogue of the LAST decoder layer (32 channels, one cout block per workgroup): the
// fine 1x1 head (descriptor rows 0..31 on MFMA + the uncertainty row as a dot product), L2
// normalisation, confidence = sigmoid(-x), float32 HWC record.  The layer's own fp16 output is then
// never written: nothing else reads it.
struct FusedHead {
  const half8* wfrag;    // [2 k-steps][64 lanes]: head rows 0..31 with columns in accumulator-register order
  const float* conf_w;   // [32] uncertainty-row weights (fp16-rounded), channel order
  const float* bias;     // [33]
  float* out[PXT_UNET_MAX_BATCH];
  int normalize[PXT_UNET_MAX_BATCH];
  int cstride, enabled;
};

// Optional fused FIRST layer (conv 3 -> 64 of the pyramid) in front of the 64 -> 64 layer: the workgroup computes the
// first layer's output for its own halo from the raw image window (normalisation, mask, fp16 hi / lo split MFMAs:
// conv_first_kernel's arithmetic) straight into its two halo buffers - the 39 MB activation per image is neither
// written nor read back, and the workgroup's prologue is a 5-KB load instead of a 41-KB one.
struct FusedFirst {
  const void* image[PXT_UNET_MAX_BATCH];   // HWC, 0..255, float or u8
  const uint8_t* mask[PXT_UNET_MAX_BATCH];
  int is_u8[PXT_UNET_MAX_BATCH];
  const float* w;   // [64][27] fp32, k = (ky * 3 + kx) * 3 + c
  const float* b;   // [64]
  int enabled;
};

// Optional per-tile skip (round 5): a tile whose whole dependency cone - through every layer up to this one - lies in a
// region where the network's INPUT is constant (the masked-out part of the query: pixloc_tracker_r9.py:224-225; the
// background of the NeRF reference render) and inside the image (no zero padding seen) has the same output vector at
// every pixel.  `flags[blockIdx.x]` == 0 marks such a tile (computed conservatively from the mask / the image,
// pxt_unet.hip); the workgroup then stores `value` (the layer's output for that constant input, obtained once by
// running THIS kernel configuration on a constant map: bit for bit what the tile would have computed) and leaves.
struct TileSkip {
  const uint8_t* flags;  // [gridDim.x] or nullptr
  const half_t* value;   // [Cout]
};

template <int TH, int BNC, int NT>
__device__ __forceinline__ void conv_fill_constant_tile(const half_t* __restrict__ value, half_t* __restrict__ out_img,
                                                        half_t* __restrict__ pool_img, int H, int W, int Cout, int ty0, int tx0,
                                                        int co0) {
  constexpr int CH = BNC / 8;  // 16-B pieces per pixel of this workgroup's channel block
  const half8* cv = (const half8*)(value + co0);
  for (int i = threadIdx.x; i < TH * 16 * CH; i += NT) {
    const int c = i % CH, p = i / CH;
    const int y = ty0 + (p >> 4), x = tx0 + (p & 15);
    if (y < H && x < W) *(half8*)(out_img + ((size_t)y * W + x) * Cout + co0 + 8 * c) = cv[c];
  }
  if (pool_img) {
    const int Hp = H >> 1, Wp = W >> 1;
    for (int i = threadIdx.x; i < (TH / 2) * 8 * CH; i += NT) {
      const int c = i % CH, p = i / CH;
      const int y = (ty0 >> 1) + (p >> 3), x = (tx0 >> 1) + (p & 7);
      if (y < Hp && x < Wp) *(half8*)(pool_img + ((size_t)y * Wp + x) * Cout + co0 + 8 * c) = cv[c];
    }
  }
}

struct ConvArgs {
  const half_t* in;      // [n_img][H][W][Cin]  (UPCAT: the skip tensor [n_img][Hs][Ws][Cin - Cp])
  int H, W, Cin;
  const half_t* wpk;     // packed taps, see pack_conv_weights()
  const float* bias;
  int Cout, relu;
  half_t* out;           // [n_img][H][W][Cout]
  float* partial;        // split-K slabs [z][n_img][H][W][Cout] (gridDim.z > 1)
  UpSrc up;
  half_t* pool;          // optional [n_img][H/2][W/2][Cout]: 2x2 max-pool of `out` (gridDim.z == 1 only)
  FusedHead head;        // head.enabled: Cout == 32, gridDim.z == 1
  FusedFirst first;      // first.enabled: Cin == 64, gridDim.z == 1, the FIRST kernel variant
  TileSkip skip;         // skip.flags: gridDim.z == 1, no fused head
};

// Host-side mirror of the packed layout: element (cout, tap, cin) lives at
//   ((((cin / 32) * 9 + tap) * 2 + s) * (Cout / 32) + cout / 32) * 512 + lane * 8 + j
// with k = cin % 32, s = k / 16, lane = (cout % 32) + 32 * ((k % 16) / 8), j = k % 8.
__host__ __device__ inline size_t packed_weight_index(int cout, int tap, int cin, int Cout) {
  const int k = cin & 31, s = k >> 4, lane = (cout & 31) + 32 * ((k & 15) >> 3), j = k & 7;
  return ((((size_t)(cin >> 5) * 9 + tap) * 2 + s) * (size_t)(Cout >> 5) + (cout >> 5)) * 512 + lane * 8 + j;
}

__global__ void pack_conv_weights_kernel(const half_t* __restrict__ w /* [Cout][9][Cin] */, int Cin, int Cout,
                                         half_t* __restrict__ packed) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Cout * 9 * Cin) return;
  const int cin = (int)(i % Cin);
  const int tap = (int)((i / Cin) % 9);
  const int cout = (int)(i / ((long long)Cin * 9));
  packed[packed_weight_index(cout, tap, cin, Cout)] = w[i];
}

typedef __attribute__((ext_vector_type(2))) _Float16 half2v;

// Returns x through an empty asm: the compiler can no longer prove the value loop-invariant, so the
// LDS address arithmetic derived from it is recomputed where it is used (a few VALU ops per chunk)
// instead of being hoisted out of the chunk loop into ~30 long-lived registers.
// Compile-time loop: the 18 (tap, k-step) steps of a chunk MUST be straight-line code (the register
// rings are indexed by the step); `#pragma unroll` silently gave up on the largest variant.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ inline int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// 2x2 max over (row pair = lanes l, l ^ 16) x (column pair = lanes l, l ^ 1) of two packed fp16 pairs,
// without touching LDS: v_permlane16_swap exchanges the odd 16-lane rows of one register with the even
// rows of another (a register swapped with its own copy leaves both row partners side by side), the
// column partner comes through a DPP quad permutation.
__device__ inline unsigned pool2x2_pk(unsigned v) {
  auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  half2v a = __builtin_bit_cast(half2v, (unsigned)sw[0]), b = __builtin_bit_cast(half2v, (unsigned)sw[1]);
  half2v m = __builtin_elementwise_max(a, b);
  const unsigned mu = __builtin_bit_cast(unsigned, m);
  const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)mu, (int)mu, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
  m = __builtin_elementwise_max(m, __builtin_bit_cast(half2v, nb));
  return __builtin_bit_cast(unsigned, m);
}

// (launch bounds: the UPCAT variants ask for ONE wave per SIMD only so that the register allocator
// does not spill -- with the 256-register cap it schedules itself into 51 spills, uncapped it needs
// 228 registers, which still runs two waves per SIMD.)
// AR = depth of the filter-fragment ring (fragments travel AR - 1 steps ahead).  3 covers an L2 hit
// when two or three waves share a SIMD; the deep layers (60x80 and 30x40 maps: <= 1 workgroup per CU,
// 2.4-4.7 MB of taps per layer that every workgroup streams once, cold) use 9 with one wave per SIMD:
// their loop was bound by the miss latency of each new fragment, not by the matrix pipe.
// Timing experiment (scripts/conv_stamps.py, a library built with -DPXT_EXP_STAMPS=1): s_memtime stamps of
// every workgroup's first lane at kernel start (0), after the first halo chunk (1), after each 32-channel
// chunk (4 + i), after the loop (2) and after the epilogue (3); s_memrealtime at 14 / 15.
#if PXT_EXP_STAMPS
__device__ unsigned long long pxt_stamps[8192 * 16];
#define PXT_STAMP(k)                                                                              \
  do {                                                                                            \
    if (threadIdx.x == 0) {                                                                       \
      const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);        \
      if (wg_ < 8192) pxt_stamps[wg_ * 16 + (k)] = __builtin_amdgcn_s_memtime();                 \
      if (wg_ < 8192 && ((k) == 0 || (k) == 3))                                                   \
        pxt_stamps[wg_ * 16 + ((k) == 0 ? 14 : 15)] = __builtin_amdgcn_s_memrealtime();          \
    }                                                                                             \
  } while (0)
#else
#define PXT_STAMP(k)
#endif
template <int CW, int PBW, int WC, int WP, bool UPCAT, int AR = 3, bool FIRST = false>
__global__ __launch_bounds__(256, AR > 3 ? 1 : 2) void conv3x3_v2_kernel(const ConvArgs a) {
  static_assert(!FIRST || !UPCAT, "the fused first layer feeds a plain layer");
  static_assert(18 % AR == 0, "the ring is indexed by the step modulo AR at compile time");
  static_assert(WC * WP == 4, "four waves per workgroup");
  constexpr int TH = 2 * PBW * WP, HR = TH + 2;
  constexpr int BNC = 32 * CW * WC;
  constexpr int kElems = HR * kV2Cols * 4;      // 16-B pieces of one halo chunk
  constexpr int KH = (kElems + 255) / 256;      // pieces per thread
  constexpr int NB = 2;                         // staged in two batches to bound the live registers
  constexpr int KB = (KH + NB - 1) / NB;
  constexpr int kBuf = HR * kV2RowBytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  PXT_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WP, wp = wave % WP;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int tiles_x = (W + 15) >> 4;
  const int tiles_per = tiles_x * ((H + TH - 1) / TH);
  const int img = blockIdx.x / tiles_per, tile = blockIdx.x % tiles_per;
  const int n_img = gridDim.x / tiles_per;
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * 16;
  const int co0 = blockIdx.y * BNC;
  const int Cs = UPCAT ? Cin - a.up.Cp : Cin;
  const int in_w = UPCAT ? a.up.Ws : W;
  const half_t* in = a.in + (size_t)img * (UPCAT ? a.up.Hs : H) * in_w * Cs;
  const half_t* prev = UPCAT ? a.up.prev + (size_t)img * a.up.Hp * a.up.Wp * a.up.Cp : nullptr;
  if (a.skip.flags != nullptr && a.skip.flags[blockIdx.x] == 0) {  // (workgroup-uniform) a constant tile: fill and leave
    conv_fill_constant_tile<TH, BNC, 256>(a.skip.value, a.out + (size_t)img * H * W * Cout,
                                          a.pool ? a.pool + (size_t)img * (H >> 1) * (W >> 1) * Cout : nullptr, H, W, Cout, ty0,
                                          tx0, co0);
    return;
  }

  // the workgroup's biases wait in LDS for the epilogue (fetched there from global memory they cost the
  // epilogue ~2.5k cycles of exposed latency; stamps)
  float* const s_bias = (float*)(smem + 2 * kBuf + (UPCAT ? (PBW * WP + 2) * 10 * 64 : 0));
  if (tid < BNC) s_bias[tid] = a.bias[co0 + tid];

  f32x16 acc[CW][PBW];
#pragma unroll
  for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int p = 0; p < PBW; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.f;

  const int r31 = lane & 31, khalf = lane >> 5;
  // B-fragment read offsets of this lane for the 3 horizontal taps x 2 k-steps; the vertical tap
  // and the pixel block are immediates ((2 * pb + ky) rows).
  int boff[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int col = (r31 & 15) + kx;
      boff[kx][s] = (2 * PBW * wp + (r31 >> 4)) * kV2RowBytes + col * 64 + (((2 * s + khalf) ^ ((col >> 2) & 3)) << 4);
    }

  // K range of this workgroup (split-K over gridDim.z)
  const int n_chunks = Cin >> 5;
  const int per_z = (n_chunks + (int)gridDim.z - 1) / (int)gridDim.z;
  const int ch_begin = (int)blockIdx.z * per_z, ch_end = min(n_chunks, ch_begin + per_z);

  // A fragments: packed [chunk][tap][s][cout block][lane][8]; this wave's CW blocks are adjacent
  const size_t a_step = (size_t)(Cout >> 5) * 512;  // halves between consecutive (tap, s) steps
  const half_t* aptr = a.wpk + ((size_t)ch_begin * 18) * a_step + (size_t)((co0 >> 5) + CW * wc) * 512 + lane * 8;

  // ---- halo staging -------------------------------------------------------------------------
  // Every staging load is UNCONDITIONAL (bounds-checked buffer loads that return zeros outside the
  // image, clamped coordinates for the low-resolution patch): a load behind a lane-divergent branch
  // makes hipcc's vmcnt bookkeeping conservative, and the filter-fragment waits then drain the halo
  // loads right after they were issued.
  half8 r_in[KB];
  unsigned goff[KH];  // byte offset of this thread's pieces in `in` (2^31 = outside the image -> zeros)
#pragma unroll
  for (int k = 0; k < KH; ++k) {
    const int i = tid + 256 * k;
    const int pix = i >> 2, seg = i & 3;
    const int gy = ty0 + pix / kV2Cols - 1, gx = tx0 + pix % kV2Cols - 1;
    const bool ok = i < kElems && gy >= 0 && gy < H && gx >= 0 && gx < W;
    goff[k] = ok ? ((unsigned)(gy * in_w + gx) * (unsigned)Cs + (unsigned)(seg * 8)) * 2u : 0x80000000u;
  }
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)in, 0, (int)((size_t)(UPCAT ? a.up.Hs : H) * in_w * Cs * 2), 0x00020000);
  const int cp0 = UPCAT ? a.up.Cp : 0;  // channels [0, cp0) of the conv input come from `prev`
  auto halo_issue = [&](int c0, int batch) {
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const int k = batch * KB + kk;
      if (k < KH)
        r_in[kk] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(
                                                 in_rsrc, (int)(goff[k] + (unsigned)((c0 - cp0) * 2)), 0, 0));
    }
  };
  auto halo_write = [&](int buf, int batch) {
    const int t_ = opaque(tid);
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const int k = batch * KB + kk;
      const int i = t_ + 256 * k;
      const int pix = i >> 2, seg = i & 3;
      const int hy = pix / kV2Cols, hx = pix % kV2Cols;
      if (k < KH && i < kElems)
        *(half8*)(smem + buf * kBuf + hy * kV2RowBytes + hx * 64 + ((seg ^ ((hx >> 2) & 3)) << 4)) = r_in[kk];
    }
  };

  // UPCAT, channels below Cp: the low-resolution patch under this tile's halo ((TH/2 + 2) x 10 pixels
  // of `prev`, one chunk) is copied raw into LDS, then every thread forms its halo pieces from it with
  // the bilinear x2 weights (exactly 0, 1/4, 3/4; align_corners = False) -- LDS latency instead of
  // four dependent global loads per piece, and each low-resolution pixel is fetched once.
  constexpr int PH = TH / 2 + 2, PW = 10;
  constexpr int kPatchElems = PH * PW * 4, KP = (kPatchElems + 255) / 256;
  static_assert(!UPCAT || KP <= KB, "the patch pieces reuse the halo staging registers");
  char* const patch = smem + 2 * kBuf;
  unsigned poff[UPCAT ? KP : 1], up_src[UPCAT ? KH : 1];
  if (UPCAT) {
    const int py0 = (ty0 >> 1) - 1, px0 = (tx0 >> 1) - 1;
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const int i = min(tid + 256 * kp, kPatchElems - 1);
      const int pp = i >> 2, seg = i & 3;
      const int sy = min(max(py0 + pp / PW, 0), a.up.Hp - 1), sx = min(max(px0 + pp % PW, 0), a.up.Wp - 1);
      poff[kp] = (unsigned)(sy * a.up.Wp + sx) * (unsigned)a.up.Cp + (unsigned)(seg * 8);
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int i = tid + 256 * k;
      const int pix = i >> 2, seg = i & 3;
      const int gy = ty0 + pix / kV2Cols - 1, gx = tx0 + pix % kV2Cols - 1;
      const bool ok = i < kElems && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float sy = fmaxf(((float)gy + 0.5f) * 0.5f - 0.5f, 0.f);
      const float sx = fmaxf(((float)gx + 0.5f) * 0.5f - 0.5f, 0.f);
      const int y0 = min((int)sy, a.up.Hp - 1), x0 = min((int)sx, a.up.Wp - 1);
      const int y1 = min(y0 + 1, a.up.Hp - 1), x1 = min(x0 + 1, a.up.Wp - 1);
      const float ay = sy - (float)y0, ax = sx - (float)x0;  // exactly 0, 0.25 or 0.75 inside the image
      const int pidx = ok ? (y0 - py0) * PW + (x0 - px0) : 0;
      up_src[k] = ((unsigned)(pidx * 64 + seg * 16) << 8) | (ok ? 64u : 0u) | (x1 != x0 ? 1u : 0u) | (y1 != y0 ? 2u : 0u) |
                  (ax == 0.25f ? 4u : ax == 0.75f ? 8u : 0u) | (ay == 0.25f ? 16u : ay == 0.75f ? 32u : 0u);
    }
  }
  auto patch_issue = [&](int c0) {
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) r_in[kp] = *(const half8*)(prev + (poff[UPCAT ? kp : 0] + (unsigned)c0));
  };
  auto patch_write = [&]() {
    const int t_ = opaque(tid);
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const int i = t_ + 256 * kp;
      if (i < kPatchElems) *(half8*)(patch + i * 16) = r_in[kp];
    }
  };
  auto patch_interp = [&](int buf) {
    const int t_ = opaque(tid);
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int i = t_ + 256 * k;
      const int pix = i >> 2, seg = i & 3;
      const int hy = pix / kV2Cols, hx = pix % kV2Cols;
      const unsigned f = (unsigned)opaque((int)up_src[UPCAT ? k : 0]);  // (not hoisted: the weights are re-derived per chunk)
      const char* src = patch + (f >> 8);
      const int dx = (f & 1u) ? 64 : 0, dy = (f & 2u) ? PW * 64 : 0;
      const float ax = (f & 4u) ? 0.25f : (f & 8u) ? 0.75f : 0.f;
      const float ay = (f & 16u) ? 0.25f : (f & 32u) ? 0.75f : 0.f;
      const half8 p00 = *(const half8*)(src), p01 = *(const half8*)(src + dx);
      const half8 p10 = *(const half8*)(src + dy), p11 = *(const half8*)(src + dy + dx);
      // the four bilinear weights are products of {0, 1/4, 3/4, 1}: exact in fp16.  Packed-fp16 multiply-adds (one
      // v_pk_mul + three v_pk_fma per channel pair): the fp32 version of this blend (4 conversions in, 6 FMAs, one
      // conversion out per channel = ~90 VALU instructions per piece) cost the decoder layers as much issue time as
      // their MFMAs (round-3 timeline: 86 us per 640x480 image for the last layer's 22.6 GFLOP).
      const bool ok_ = (f & 64u) != 0;
      const half_t wa = (half_t)(ok_ ? (1.f - ax) * (1.f - ay) : 0.f), wb = (half_t)(ok_ ? ax * (1.f - ay) : 0.f);
      const half_t wc = (half_t)(ok_ ? (1.f - ax) * ay : 0.f), wd = (half_t)(ok_ ? ax * ay : 0.f);
      half8 v = p00 * wa;
      v = p01 * wb + v;
      v = p10 * wc + v;
      v = p11 * wd + v;
      if (i < kElems)
        *(half8*)(smem + buf * kBuf + hy * kV2RowBytes + hx * 64 + ((seg ^ ((hx >> 2) & 3)) << 4)) = v;
      __builtin_amdgcn_sched_barrier(0);  // one piece's four taps live at a time (register budget)
    }
  };

  // A K range of exactly TWO chunks (the 64-input-channel layers at 640x480 and 320x240): both halos are fetched in
  // the prologue, back to back, so the workgroup pays ONE memory latency.  (Staged the usual way, chunk 1's loads had
  // the five MFMA steps of chunk 0 - ~650 cycles - to arrive before their LDS write needed them; stamps: prologue 7.7 k
  // + chunk 0 8.1 k + chunk 1 4.9 k cycles against 2.3 k of MFMA issue per chunk.)
  const bool preload2 = FIRST || (!UPCAT && (ch_end - ch_begin) == 2);
  if constexpr (FIRST) {
    // ---- the pyramid's first layer, computed for this tile's halo ((TH + 2) x 18 pixels x 64 channels = both chunks)
    constexpr int FW = 20;  // raw window: halo + 1 pixel on every side
    float* const s_px = (float*)(smem + 2 * kBuf + 32 * CW * WC * 4);
    {
      const void* image = a.first.image[img];
      const uint8_t* mask = a.first.mask[img];
      const bool u8 = a.first.is_u8[img] != 0;
      const float mean[3] = {0.485f, 0.456f, 0.406f};
      const float istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
      for (int i = tid; i < (HR + 2) * FW; i += 256) {
        const int yy = ty0 + i / FW - 2, xx = tx0 + i % FW - 2;
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
    }
    half8 wh[2][2], wl[2][2];  // A operand: row = first-layer channel 32 cb + r31, k = 16 s + 8 khalf + j
    int koff[2][8];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int k = 16 * s2 + 8 * khalf + jj;  // (ky, kx, c) = (k / 9, (k / 3) % 3, k % 3)
        koff[s2][jj] = k < 27 ? ((k / 9) * FW + (k / 3) % 3) * 3 + k % 3 : -1;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float w = k < 27 ? a.first.w[(size_t)(32 * cb + r31) * 27 + k] : 0.f;
          const half_t h = (half_t)w;
          wh[cb][s2][jj] = h;
          wl[cb][s2][jj] = (half_t)(w - (float)h);
        }
      }
    __syncthreads();
    constexpr int kHaloPix = HR * kV2Cols, kBlocks = (kHaloPix + 31) / 32;
    for (int bi = wave; bi < kBlocks; bi += 4) {
      const int q = 32 * bi + r31;
      const bool live = q < kHaloPix;
      const int qq = live ? q : 0;
      const int hy = qq / kV2Cols, hx = qq % kV2Cols;
      const float* win = s_px + (hy * FW + hx) * 3;  // the 3x3 window of halo pixel (hy, hx) starts one pixel up-left of it
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
      // the layer's zero padding: halo pixels outside the image are zeros, not the first layer's response to padding
      const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
      const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        f32x16 fa;
#pragma unroll
        for (int r = 0; r < 16; ++r) fa[r] = a.first.b[32 * cb + (r & 3) + 8 * (r >> 2) + 4 * khalf];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          fa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cb][s2], xh[s2], fa, 0, 0, 0);
          fa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cb][s2], xh[s2], fa, 0, 0, 0);
          fa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cb][s2], xl[s2], fa, 0, 0, 0);
        }
        // chunk cb = channels 32 cb .. + 31: piece g holds channels 8 g .. + 7, this lane its half 4 khalf .. + 3
        char* rec = smem + cb * kBuf + hy * kV2RowBytes + hx * 64 + khalf * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = inside ? (half_t)fmaxf(fa[4 * g + j], 0.f) : (half_t)0.f;
          if (live) *(half4*)(rec + ((g ^ ((hx >> 2) & 3)) << 4)) = o;
        }
      }
    }
  } else if (ch_begin < ch_end) {  // first chunk: staged synchronously
    if (UPCAT && ch_begin * 32 < cp0) {
      patch_issue(ch_begin * 32);
      patch_write();
      __syncthreads();
      patch_interp(0);
    } else if (preload2) {
      half8 r_nx[KH];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        halo_issue(ch_begin * 32, b);
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
          const int k = b * KB + kk;
          if (k < KH)
            r_nx[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(
                                                    in_rsrc, (int)(goff[k] + (unsigned)(((ch_begin + 1) * 32 - cp0) * 2)), 0, 0));
        }
        halo_write(0, b);
      }
      const int t_ = opaque(tid);
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        const int i = t_ + 256 * k;
        const int pix = i >> 2, seg = i & 3;
        const int hy = pix / kV2Cols, hx = pix % kV2Cols;
        if (i < kElems) *(half8*)(smem + kBuf + hy * kV2RowBytes + hx * 64 + ((seg ^ ((hx >> 2) & 3)) << 4)) = r_nx[k];
      }
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        halo_issue(ch_begin * 32, b);
        halo_write(0, b);
      }
    }
  }

  // Explicit software pipeline: filter fragments travel AR - 1 steps ahead of their use (an L2 hit
  // is ~300-500 cycles under load, one step only 4-8 MFMAs), pixel fragments one step ahead (LDS).
  // The rings are indexed with compile-time constants (18 steps per chunk, 18 % AR == 0).
  PXT_STAMP(1);
  half8 a_q[AR][CW], b_q[2][PBW];
  const int n_steps = (ch_end - ch_begin) * 18;
  if (n_steps > 0) {
#pragma unroll
    for (int d = 0; d < AR - 1; ++d)  // (a K range holds >= 18 steps)
#pragma unroll
      for (int c = 0; c < CW; ++c) a_q[d][c] = *(const half8*)(aptr + (size_t)d * a_step + c * 512);
  }
  // address of the fragment AR - 1 steps ahead, clamped to the last one of this K range so that the
  // prefetch needs no branch at the end
  const half_t* const a_last = aptr + (size_t)max(n_steps - 1, 0) * a_step;
  aptr += (size_t)(AR - 1) * a_step;

  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const bool more = ch + 1 < ch_end && !preload2;
    const char* sbuf = smem + ((ch - ch_begin) & 1) * kBuf;
    const int nbuf = ((ch - ch_begin) & 1) ^ 1;
    __syncthreads();  // this chunk's halo is complete; everyone is done reading the other buffer
#pragma unroll
    for (int p = 0; p < PBW; ++p) b_q[0][p] = *(const half8*)(sbuf + boff[0][0] + (2 * p) * kV2RowBytes);
    static_for<0, 18>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      {
        const half_t* ap = aptr < a_last ? aptr : a_last;
#pragma unroll
        for (int c = 0; c < CW; ++c) a_q[(st + AR - 1) % AR][c] = *(const half8*)(ap + c * 512);
      }
      aptr += a_step;
      if (more) {  // stage the next chunk's halo into the other buffer (workgroup-uniform branches)
        const int c1 = (ch + 1) * 32;
        if (UPCAT && c1 < cp0) {
          if constexpr (st == 0) patch_issue(c1);
          if constexpr (st == 5) { patch_write(); __syncthreads(); }
          if constexpr (st == 7) patch_interp(nbuf);
        } else {
          if constexpr (st == 0) halo_issue(c1, 0);
          if constexpr (st == 5) { halo_write(nbuf, 0); halo_issue(c1, 1); }
          if constexpr (st == 11) halo_write(nbuf, 1);
        }
      }
      if constexpr (st < 17) {
        constexpr int tn = (st + 1) >> 1, sn = (st + 1) & 1;
#pragma unroll
        for (int p = 0; p < PBW; ++p)
          b_q[(st + 1) & 1][p] = *(const half8*)(sbuf + boff[tn % 3][sn] + (2 * p + tn / 3) * kV2RowBytes);
      }
      // hipcc's scheduler otherwise sinks the prefetch loads down to their first use (shorter live
      // ranges), which is exactly the latency this pipeline exists to hide: pin them above the MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < PBW; ++p)
#pragma unroll
        for (int c = 0; c < CW; ++c)
          acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_q[st % AR][c], b_q[st & 1][p], acc[c][p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (ch - ch_begin < 10) PXT_STAMP(4 + ch - ch_begin);
  }
  PXT_STAMP(2);

  // ---- epilogue: D[row = cout][col = pixel]; lane: col = lane & 31, rows (r&3) + 8*(r>>2) + 4*khalf
  __syncthreads();  // every wave is done with the halo buffers: the plain-output path stages tiles there
  PXT_STAMP(12);
  const int cw0 = co0 + 32 * CW * wc;
  half_t* out = a.out + (size_t)img * H * W * Cout;
#pragma unroll
  for (int p = 0; p < PBW; ++p) {
    const int gy = ty0 + 2 * PBW * wp + 2 * p + (r31 >> 4), gx = tx0 + (r31 & 15);
    const bool inside = gy < H && gx < W;
    if (gridDim.z > 1) {
      if (!inside) continue;
      float* pd = a.partial + ((((size_t)blockIdx.z * n_img + img) * H + gy) * W + gx) * Cout + cw0;
#pragma unroll
      for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(pd + 32 * c + 8 * g + 4 * khalf) =
              make_float4(acc[c][p][4 * g + 0], acc[c][p][4 * g + 1], acc[c][p][4 * g + 2], acc[c][p][4 * g + 3]);
      continue;
    }
    if (CW == 1 && a.head.enabled) {  // workgroup-uniform
      // bias + ReLU + fp16: the 16 accumulator registers ARE the two B fragments of the head's GEMM
      // (its weight columns were permuted to this register order when the context was created)
      half8 hb[2];
      float cdot = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const half_t hv = (half_t)fmaxf(acc[0][p][r] + a.bias[ch], 0.f);
        hb[r >> 3][r & 7] = hv;
        cdot += (float)hv * a.head.conf_w[ch];
      }
      f32x16 hd = {0};
      hd = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.head.wfrag[lane], hb[0], hd, 0, 0, 0);
      hd = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.head.wfrag[64 + lane], hb[1], hd, 0, 0, 0);
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hd[r] += a.head.bias[(r & 3) + 8 * (r >> 2) + 4 * khalf];
        ss += hd[r] * hd[r];
      }
      ss += __shfl_xor(ss, 32, 64);
      cdot += __shfl_xor(cdot, 32, 64);
      const float inv = a.head.normalize[img] ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 1.f;
      if (inside) {
        float* o = a.head.out[img] + ((size_t)gy * W + gx) * a.head.cstride;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(o + 8 * g + 4 * khalf) = make_float4(hd[4 * g] * inv, hd[4 * g + 1] * inv, hd[4 * g + 2] * inv, hd[4 * g + 3] * inv);
        if (khalf == 0) *(float4*)(o + 32) = make_float4(1.f / (1.f + expf(cdot + a.head.bias[32])), 0.f, 0.f, 0.f);
      }
      continue;
    }
    // Plain output.  The accumulator layout puts ONE pixel's 8 channels in a lane, so direct 16-B stores hit
    // 32 different 128-B lines per instruction with 32 B each, and the epilogue ran at the speed of the
    // address unit (stamps: ~9.8k cycles, a quarter of a 128-channel layer's workgroup time).  The tile goes
    // through LDS instead (the halo buffers, free after the loop): written pixel-major with a padded pitch,
    // read back so that 4 * CW consecutive lanes cover one pixel's 64 * CW contiguous bytes.
    constexpr int kPitch = CW * 64 + 16;
    char* const stage = smem + wave * (32 * kPitch);
    half_t* pdst = nullptr;
    const bool pool_lane = a.pool && (r31 & 17) == 0 && (gy >> 1) < (H >> 1) && (gx >> 1) < (W >> 1);
    if (a.pool)
      pdst = a.pool + (((size_t)img * (H >> 1) + (gy >> 1)) * (W >> 1) + (gx >> 1)) * Cout + cw0;
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned lo[2], hi[2];  // packed fp16 pairs of row groups g and g + 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 bv = *(const float4*)(s_bias + 32 * (CW * wc + c) + 8 * (g + h) + 4 * khalf);
          float v0 = acc[c][p][4 * (g + h) + 0] + bv.x, v1 = acc[c][p][4 * (g + h) + 1] + bv.y;
          float v2 = acc[c][p][4 * (g + h) + 2] + bv.z, v3 = acc[c][p][4 * (g + h) + 3] + bv.w;
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          half4 o;
          o[0] = (half_t)v0; o[1] = (half_t)v1; o[2] = (half_t)v2; o[3] = (half_t)v3;
          if (a.pool) {  // wave-uniform branch
            const uint2 m = make_uint2(pool2x2_pk(((const unsigned*)&o)[0]), pool2x2_pk(((const unsigned*)&o)[1]));
            if (pool_lane) *(uint2*)(pdst + 32 * c + 8 * (g + h) + 4 * khalf) = m;
          }
          (h == 0 ? lo : hi)[0] = ((const unsigned*)&o)[0];
          (h == 0 ? lo : hi)[1] = ((const unsigned*)&o)[1];
        }
        // lanes 0-31 (khalf 0) end up with channels 8g .. 8g+7, lanes 32-63 with 8(g+1) .. 8(g+1)+7
        auto s0 = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
        *(uint4*)(stage + r31 * kPitch + (32 * c + 8 * (g + khalf)) * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    if (p == 0) PXT_STAMP(13);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the staged tile is read by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int kLpp = 4 * CW;  // lanes per pixel (16 B each)
#pragma unroll
    for (int it = 0; it < 2 * CW; ++it) {
      const int pix = it * (64 / kLpp) + lane / kLpp, piece = lane % kLpp;
      const uint4 v = *(const uint4*)(stage + pix * kPitch + piece * 16);
      const int oy = ty0 + 2 * PBW * wp + 2 * p + (pix >> 4), ox = tx0 + (pix & 15);
      if (oy < H && ox < W) *(uint4*)(out + ((size_t)oy * W + ox) * Cout + cw0 + piece * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();  // the next pixel block overwrites the stage
  }
  PXT_STAMP(3);
}

}  // namespace pxt
