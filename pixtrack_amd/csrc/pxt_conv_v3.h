// 3x3 convolution (pad 1), NHWC fp16 -> NHWC fp16, MFMA implicit GEMM -- third generation (round 3).
// Included by pxt_unet.hip after pxt_conv_v2.h (shares ConvArgs, static_for, opaque, PXT_STAMP).
//
// What the second kernel's counters and timelines said: its loop ran the matrix pipe at ~66 %, the
// missing third being the filter fragments every wave fetched for itself straight from L2 (with PBW = 2 tiles that
// stream alone is 64 B/clk/CU = the whole L1 path; a wave that cannot issue its load cannot issue its MFMAs either),
// and every MFMA read a fresh pixel fragment from LDS.  This kernel changes the dataflow of a (k-step, kx) GROUP:
//  * PIXEL FRAGMENTS ARE SHARED BY THE THREE VERTICAL TAPS.  A wave's 32-pixel block p is {row p, row p + PBW} x 16
//    columns of its 2*PBW-row strip, so the fragment block p needs for tap ky is "rows (p + ky, p + ky + PBW) at column
//    shift kx" = fragment index p + ky: a group reads PBW + 2 fragments for its 3 * PBW * CW MFMAs instead of 3 * PBW.
//  * THE FILTER FRAGMENTS OF A GROUP GO THROUGH LDS, ONCE PER WORKGROUP.  They are packed [chunk][s][kx][ky][cout
//    block][lane][8], so a group's taps for the workgroup's cout blocks are three contiguous runs; the workgroup
//    copies them (plain 16-B loads one group ahead, ds_write at the end of the group, double buffered) and every wave
//    reads its fragments with conflict-free ds_read_b128: 1-3 vector-memory instructions per thread and group instead
//    of 3 * CW per wave, the L1 path is left to the halo.
//  * one barrier per group; two workgroups per CU (<= 80 KB of LDS, <= 256 registers) fill each other's barrier and
//    epilogue gaps.
//  * KC = 16-channel chunks (32-B pixel records) halve the halo buffers for the 64- / 32-channel layers whose
//    workgroup tile is 32 rows high.
#pragma once

namespace pxt {

template <int KC>
struct V3Geo {
  static constexpr int kPix = KC * 2;               // bytes of a pixel record in LDS
  static constexpr int kNP = KC / 8;                // 16-B pieces per pixel
  static constexpr int kRow = 20 * kPix;            // row pitch: 18 records used
  static constexpr int kPixPerBankRow = 256 / kPix; // pixels per 256-B bank row
  __device__ static inline int swz(int col) {       // piece-index XOR that spreads 16 consecutive columns over 16 slots
    return kPixPerBankRow == 4 ? ((col >> 2) & 3) : ((col >> 3) & 1);
  }
};

// packed index of element (cout, tap = ky * 3 + kx, cin) in the v3 layout
__host__ __device__ inline size_t packed_weight_index_v3(int cout, int tap, int cin, int Cout) {
  const int ky = tap / 3, kx = tap % 3;
  const int k = cin & 31, s = k >> 4, lane = (cout & 31) + 32 * ((k & 15) >> 3), j = k & 7;
  const size_t group = (size_t)(cin >> 5) * 6 + s * 3 + kx;
  return ((group * 3 + ky) * (size_t)(Cout >> 5) + (cout >> 5)) * 512 + lane * 8 + j;
}

__global__ void pack_conv_weights_v3_kernel(const half_t* __restrict__ w /* [Cout][9][Cin] */, int Cin, int Cout,
                                            half_t* __restrict__ packed) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Cout * 9 * Cin) return;
  const int cin = (int)(i % Cin);
  const int tap = (int)((i / Cin) % 9);
  const int cout = (int)(i / ((long long)Cin * 9));
  packed[packed_weight_index_v3(cout, tap, cin, Cout)] = w[i];
}

// packed fp16 pair max
__device__ inline unsigned pk_max(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(half2v, a), __builtin_bit_cast(half2v, b)));
}

constexpr int v3_lds_bytes(int CW, int PBW, int WC, int WP, int KC, bool upcat = false, int KS = 1) {
  return KS * (2 * (2 * PBW * WP + 2) * (20 * KC * 2) + 2 * 3 * CW * WC * 1024) + 32 * CW * WC * 4 +
         (upcat ? (PBW * WP + 2) * 10 * KC * 2 : 0);
}

// UPCAT (decoder layers): the conv input is concat(bilinear x2 upsample of `prev`, skip); channels below Cp are formed
// in the staging from the low-resolution patch under the tile's halo, exactly as in the second kernel.
// KS = 2 (the layers whose grid leaves one workgroup per CU): the workgroup has EIGHT waves - two quartets that walk the
// two halves of the K range with their own halo / filter buffers, side by side on the SIMDs (two waves per SIMD where
// the four-wave workgroup left one), and add their accumulators through LDS before the epilogue.
template <int CW, int PBW, int WC, int WP, int KC, bool UPCAT = false, int KS = 1>
__global__ __launch_bounds__(256 * KS, 2) void conv3x3_v3_kernel(const ConvArgs a) {
  static_assert(WC * WP == 4, "four waves per quartet");
  static_assert(KS == 1 || (KS == 2 && !UPCAT), "K split across two wave quartets: plain layers only");
  static_assert(KC == 16 || KC == 32, "chunk of 16 or 32 input channels");
  using G = V3Geo<KC>;
  constexpr int TH = 2 * PBW * WP, HR = TH + 2;
  constexpr int NCB = CW * WC, BNC = 32 * NCB;
  constexpr int NG = 3 * (KC / 16);                 // (k-step, kx) groups per chunk
  constexpr int kBuf = HR * G::kRow;
  constexpr int kABuf = 3 * NCB * 1024;
  constexpr int kElems = HR * kV2Cols * G::kNP;     // 16-B pieces of one halo chunk
  constexpr int KH = (kElems + 255) / 256;
  constexpr int kAElems = 3 * NCB * 64;             // 16-B pieces of one group's filter fragments
  constexpr int KA = (kAElems + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kQuartet = 2 * kBuf + 2 * kABuf;  // LDS of one wave quartet
  const int kh = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);  // which half of the K range
  char* const s_halo = smem + kh * kQuartet;
  char* const s_a = s_halo + 2 * kBuf;
  float* const s_bias = (float*)(smem + KS * kQuartet);
  char* const patch = smem + KS * kQuartet + BNC * 4;  // UPCAT: the low-resolution patch of one chunk

  PXT_STAMP(0);
  const int tid = threadIdx.x & 255;  // index inside the quartet
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
  const int Cs = UPCAT ? Cin - a.up.Cp : Cin;  // channels of the tensor behind `in` (UPCAT: the skip tensor)
  const int in_w = UPCAT ? a.up.Ws : W;
  const int cp0 = UPCAT ? a.up.Cp : 0;         // channels [0, cp0) of the conv input come from `prev`
  const half_t* in = a.in + (size_t)img * (UPCAT ? a.up.Hs : H) * in_w * Cs;
  const half_t* prev = UPCAT ? a.up.prev + (size_t)img * a.up.Hp * a.up.Wp * a.up.Cp : nullptr;
  if (a.skip.flags != nullptr && a.skip.flags[blockIdx.x] == 0) {  // (workgroup-uniform) a constant tile: pxt_conv_v2.h TileSkip
    conv_fill_constant_tile<TH, BNC, 256 * KS>(a.skip.value, a.out + (size_t)img * H * W * Cout,
                                               a.pool ? a.pool + (size_t)img * (H >> 1) * (W >> 1) * Cout : nullptr, H, W, Cout,
                                               ty0, tx0, co0);
    return;
  }

  if ((int)threadIdx.x < BNC) s_bias[threadIdx.x] = a.bias[co0 + threadIdx.x];

  f32x16 acc[CW][PBW];
#pragma unroll
  for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int p = 0; p < PBW; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.f;

  const int r31 = lane & 31, khalf = lane >> 5;
  // this lane's pixel of a block: halo row (block index j added as an immediate) and column
  const int lrow = 2 * PBW * wp + (r31 >> 4) * PBW, lcol = r31 & 15;

  // K range of this workgroup (split-K over gridDim.z), in chunks of KC channels
  const int n_chunks = Cin / KC;
  const int per_z = (n_chunks + (int)gridDim.z - 1) / (int)gridDim.z;
  int ch_begin = (int)blockIdx.z * per_z, ch_end = min(n_chunks, ch_begin + per_z);
  if (KS == 2) {  // (the host only picks KS = 2 for an even number of chunks: both quartets meet at the same barriers)
    const int half = (ch_end - ch_begin) >> 1;
    ch_begin += kh * half;
    ch_end = ch_begin + half;
  }
  const int n_groups = (ch_end - ch_begin) * NG;

  // ---- filter staging: group gg (global index) = three runs of NCB KiB --------------------------
  const size_t a_group = (size_t)3 * (Cout >> 5) * 512;  // halves per group
  const half_t* const wbase = a.wpk + (size_t)ch_begin * NG * a_group + (size_t)(co0 >> 5) * 512;
  unsigned a_src[KA];  // halves offset inside a group
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    const int i = min(tid + 256 * k, kAElems - 1);
    const int ky = i / (NCB * 64), rem = i % (NCB * 64);
    a_src[k] = (unsigned)(ky * (Cout >> 5) * 512 + rem * 8);
  }
  half8 r_a[KA];
  const half_t* a_next = wbase;  // group whose fragments the next a_issue fetches
  auto a_issue_piece = [&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    r_a[k] = *(const half8*)(a_next + a_src[k]);
  };
  auto a_write_piece = [&](auto k_c, int buf) {
    constexpr int k = decltype(k_c)::value;
    const int i = opaque(tid) + 256 * k;
    if (i < kAElems) *(half8*)(s_a + buf * kABuf + i * 16) = r_a[k];
  };

  // ---- halo staging (as in the second kernel: unconditional bounds-checked buffer loads) ----------
  half8 r_in[KH];
  unsigned goff[KH];
#pragma unroll
  for (int k = 0; k < KH; ++k) {
    const int i = tid + 256 * k;
    const int pix = i / G::kNP, seg = i % G::kNP;
    const int gy = ty0 + pix / kV2Cols - 1, gx = tx0 + pix % kV2Cols - 1;
    const bool ok = i < kElems && gy >= 0 && gy < H && gx >= 0 && gx < W;
    goff[k] = ok ? ((unsigned)(gy * in_w + gx) * (unsigned)Cs + (unsigned)(seg * 8)) * 2u : 0x80000000u;
  }
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)in, 0, (int)((size_t)(UPCAT ? a.up.Hs : H) * in_w * Cs * 2), 0x00020000);
  auto halo_issue_piece = [&](auto k_c, int c0) {
    constexpr int k = decltype(k_c)::value;
    r_in[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(goff[k] + (unsigned)((c0 - cp0) * 2)), 0, 0));
  };
  auto halo_write_piece = [&](auto k_c, int buf) {
    constexpr int k = decltype(k_c)::value;
    const int i = opaque(tid) + 256 * k;
    const int pix = i / G::kNP, seg = i % G::kNP;
    const int hy = pix / kV2Cols, hx = pix % kV2Cols;
    if (i < kElems) *(half8*)(s_halo + buf * kBuf + hy * G::kRow + hx * G::kPix + ((seg ^ G::swz(hx)) << 4)) = r_in[k];
  };

  // ---- UPCAT, channels below Cp: the low-resolution patch under this tile's halo ((TH/2 + 2) x 10 pixels of `prev`,
  // one chunk) is copied raw into LDS, then every thread forms its halo pieces from it with the bilinear x2 weights
  // (exactly 0, 1/4, 3/4; align_corners = False): each low-resolution pixel is fetched once.
  constexpr int PH = TH / 2 + 2, PW = 10;
  constexpr int kPatchElems = PH * PW * G::kNP, KP = (kPatchElems + 255) / 256;
  static_assert(!UPCAT || KP <= KH, "the patch pieces reuse the halo staging registers");
  unsigned poff[UPCAT ? KP : 1], up_src[UPCAT ? KH : 1];
  if constexpr (UPCAT) {
    const int py0 = (ty0 >> 1) - 1, px0 = (tx0 >> 1) - 1;
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const int i = min(tid + 256 * kp, kPatchElems - 1);
      const int pp = i / G::kNP, seg = i % G::kNP;
      const int sy = min(max(py0 + pp / PW, 0), a.up.Hp - 1), sx = min(max(px0 + pp % PW, 0), a.up.Wp - 1);
      poff[kp] = (unsigned)(sy * a.up.Wp + sx) * (unsigned)a.up.Cp + (unsigned)(seg * 8);
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int i = tid + 256 * k;
      const int pix = i / G::kNP, seg = i % G::kNP;
      const int gy = ty0 + pix / kV2Cols - 1, gx = tx0 + pix % kV2Cols - 1;
      const bool ok = i < kElems && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float sy = fmaxf(((float)gy + 0.5f) * 0.5f - 0.5f, 0.f);
      const float sx = fmaxf(((float)gx + 0.5f) * 0.5f - 0.5f, 0.f);
      const int y0 = min((int)sy, a.up.Hp - 1), x0 = min((int)sx, a.up.Wp - 1);
      const int y1 = min(y0 + 1, a.up.Hp - 1), x1 = min(x0 + 1, a.up.Wp - 1);
      const float ay = sy - (float)y0, ax = sx - (float)x0;  // exactly 0, 0.25 or 0.75 inside the image
      const int pidx = ok ? (y0 - py0) * PW + (x0 - px0) : 0;
      up_src[k] = ((unsigned)(pidx * G::kPix + seg * 16) << 8) | (ok ? 64u : 0u) | (x1 != x0 ? 1u : 0u) | (y1 != y0 ? 2u : 0u) |
                  (ax == 0.25f ? 4u : ax == 0.75f ? 8u : 0u) | (ay == 0.25f ? 16u : ay == 0.75f ? 32u : 0u);
    }
  }
  auto patch_issue_piece = [&](auto k_c, int c0) {
    constexpr int k = decltype(k_c)::value;
    if constexpr (UPCAT && k < KP) r_in[k] = *(const half8*)(prev + (poff[k] + (unsigned)c0));
  };
  auto patch_write_piece = [&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    if constexpr (UPCAT && k < KP) {
      const int i = opaque(tid) + 256 * k;
      if (i < kPatchElems) *(half8*)(patch + i * 16) = r_in[k];
    }
  };
  auto patch_interp_piece = [&](auto k_c, int buf) {
    constexpr int k = decltype(k_c)::value;
    if constexpr (UPCAT) {
      const int i = opaque(tid) + 256 * k;
      const int pix = i / G::kNP, seg = i % G::kNP;
      const int hy = pix / kV2Cols, hx = pix % kV2Cols;
      const unsigned f = (unsigned)opaque((int)up_src[k]);  // (not hoisted: the weights are re-derived per chunk)
      const char* src = patch + (f >> 8);
      const int dx = (f & 1u) ? G::kPix : 0, dy = (f & 2u) ? PW * G::kPix : 0;
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
      if (i < kElems) *(half8*)(s_halo + buf * kBuf + hy * G::kRow + hx * G::kPix + ((seg ^ G::swz(hx)) << 4)) = v;
    }
  };

  // ---- fragment reads ------------------------------------------------------------------------
  // pixel fragment j (halo rows lrow + j) of a group with horizontal tap kx and k-step s
  auto b_ptr = [&](int hb, int s, int kx) {
    const int col = lcol + kx;
    return s_halo + hb * kBuf + lrow * G::kRow + col * G::kPix + ((((KC == 32 ? 2 * s : 0) + khalf) ^ G::swz(col)) << 4);
  };
  auto a_ptr = [&](int ab) { return s_a + ab * kABuf + (CW * wc * 64 + lane) * 16; };

  // Software pipeline over the (k-step, kx) groups.  A group's 3 * PBW * CW MFMAs run as three phases (ky = 0, 1, 2);
  // its operands arrive in two halves: the HEAD (filter fragments of ky = 0, pixel fragments 0 .. PBW-1) is read during
  // the PREVIOUS group's last phase, the TAIL (ky = 1, 2; pixel fragments PBW, PBW + 1) during its own first phase -
  // into the registers the finished phases free, so 12 fragments stay live.  The workgroup barrier sits between
  // phases 1 and 2: every wave wrote the next group's filter fragments a phase earlier, nobody waits there, and the
  // reads that follow are covered by phase 2.  Staging instructions (one per MFMA) ride between the MFMAs of phases
  // 0 and 1.  The barrier is the raw s_barrier behind lgkmcnt(0) only: __syncthreads() would also drain the vector
  // loads issued a phase ago (cdna_hip_programming.md "Pipelining across barriers").
  half8 hA[CW], hB[PBW];
  if (n_groups > 0) {  // first chunk and first group: staged synchronously
    if (UPCAT && ch_begin * KC < cp0) {
      static_for<0, KP>([&](auto k) { patch_issue_piece(k, ch_begin * KC); });
      static_for<0, KA>([&](auto k) { a_issue_piece(k); });
      static_for<0, KP>([&](auto k) { patch_write_piece(k); });
      __syncthreads();
      static_for<0, KH>([&](auto k) { patch_interp_piece(k, 0); });
    } else {
      static_for<0, KH>([&](auto k) { halo_issue_piece(k, ch_begin * KC); });
      static_for<0, KA>([&](auto k) { a_issue_piece(k); });
      static_for<0, KH>([&](auto k) { halo_write_piece(k, 0); });
    }
    static_for<0, KA>([&](auto k) { a_write_piece(k, 0); });
    a_next = wbase + (size_t)min(1, n_groups - 1) * a_group;
    static_for<0, KA>([&](auto k) { a_issue_piece(k); });
    __syncthreads();
    const char* bp = b_ptr(0, 0, 0);
    const char* ap = a_ptr(0);
#pragma unroll
    for (int c = 0; c < CW; ++c) hA[c] = *(const half8*)(ap + c * 1024);
#pragma unroll
    for (int p = 0; p < PBW; ++p) hB[p] = *(const half8*)(bp + p * G::kRow);
  }
  PXT_STAMP(1);

  constexpr int NM = PBW * CW;                       // MFMAs per phase
  constexpr int TO = 2 * KA + KH;                    // staging instructions of a group (at most)
  constexpr int OPM = (TO + 2 * NM - 1) / (2 * NM);  // ... per MFMA of phases 0 and 1
  int gg = 0;  // running group index inside this workgroup's K range
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int hbuf = (ch - ch_begin) & 1;
    const int c_next = min(ch + 1, ch_end - 1) * KC;  // (the last chunk re-stages itself into the idle buffer: no branch)
    static_for<0, NG>([&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      constexpr int s = KC == 32 ? g / 3 : 0, kx = g % 3;
      constexpr int gn = (g + 1) % NG;
      constexpr int sn = KC == 32 ? gn / 3 : 0, kxn = gn % 3;
      const int abuf = gg & 1;
      // tail of this group
      half8 tA[2][CW], tB[2];
      {
        const char* bp = b_ptr(hbuf, s, kx);
        const char* ap = a_ptr(abuf);
#pragma unroll
        for (int ky = 1; ky < 3; ++ky)
#pragma unroll
          for (int c = 0; c < CW; ++c) tA[ky - 1][c] = *(const half8*)(ap + (ky * NCB + c) * 1024);
        tB[0] = *(const half8*)(bp + PBW * G::kRow);
        tB[1] = *(const half8*)(bp + (PBW + 1) * G::kRow);
      }
      __builtin_amdgcn_sched_barrier(0);
      // staging instruction number `op` of this group: write the next group's filter fragments (fetched a group ago),
      // fetch the ones after that, and move the next chunk's halo (fetch in the first group, write in the last but one)
      auto stage_op = [&](auto op_c) {
        constexpr int op = decltype(op_c)::value;
        if constexpr (op < KA) {
          a_write_piece(std::integral_constant<int, op>{}, abuf ^ 1);
        } else if constexpr (op < 2 * KA) {
          if constexpr (op == KA) a_next = wbase + (size_t)min(gg + 2, n_groups - 1) * a_group;
          a_issue_piece(std::integral_constant<int, op - KA>{});
        } else if constexpr (op < TO) {
          constexpr int k = op - 2 * KA;
          if (UPCAT && c_next < cp0) {  // (workgroup-uniform) the next chunk is upsampled from `prev`
            if constexpr (g == 0) patch_issue_piece(std::integral_constant<int, k>{}, c_next);
            if constexpr (g == 1) patch_write_piece(std::integral_constant<int, k>{});  // visible behind group 1's barrier
            if constexpr (g >= 2 && (k % (NG - 2)) == g - 2) patch_interp_piece(std::integral_constant<int, k>{}, hbuf ^ 1);
          } else {
            if constexpr (g == 0) halo_issue_piece(std::integral_constant<int, k>{}, c_next);
            if constexpr (g == NG - 2) halo_write_piece(std::integral_constant<int, k>{}, hbuf ^ 1);
          }
        }
      };
      // phases 0 and 1, one MFMA + its staging instructions at a time
      static_for<0, 2 * NM>([&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        constexpr int ky = m / NM, p = (m % NM) / CW, c = m % CW;
        const half8 av = ky == 0 ? hA[c] : tA[0][c];
        const half8 bv = (p + ky < PBW) ? hB[(p + ky < PBW) ? p + ky : 0] : tB[0];
        acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[c][p], 0, 0, 0);
        static_for<m * OPM, (m + 1) * OPM>([&](auto op_c) { stage_op(op_c); });
        __builtin_amdgcn_sched_barrier(0);
      });
      // every wave's staging writes are complete and visible behind this barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // head of the next group (the next chunk's halo buffer after a chunk's last group)
      half8 nA[CW], nB[PBW];
      {
        const char* bp = b_ptr(g == NG - 1 ? hbuf ^ 1 : hbuf, sn, kxn);
        const char* ap = a_ptr(abuf ^ 1);
#pragma unroll
        for (int c = 0; c < CW; ++c) nA[c] = *(const half8*)(ap + c * 1024);
#pragma unroll
        for (int p = 0; p < PBW; ++p) nB[p] = *(const half8*)(bp + p * G::kRow);
      }
      __builtin_amdgcn_sched_barrier(0);
      // phase 2
#pragma unroll
      for (int p = 0; p < PBW; ++p)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          const half8 bv = (p + 2 < PBW) ? hB[(p + 2 < PBW) ? p + 2 : 0] : tB[(p + 2 < PBW) ? 0 : p + 2 - PBW];
          acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tA[1][c], bv, acc[c][p], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CW; ++c) hA[c] = nA[c];
#pragma unroll
      for (int p = 0; p < PBW; ++p) hB[p] = nB[p];
      ++gg;
    });
    if (ch - ch_begin < 10) PXT_STAMP(4 + ch - ch_begin);
  }
  PXT_STAMP(2);

  if constexpr (KS == 2) {  // the second quartet hands its accumulators over through LDS and leaves
    __syncthreads();
    float* const xch = (float*)smem + (size_t)wave * (CW * PBW * 16 * 64) + lane;
    static_assert(4 * CW * PBW * 16 * 64 * 4 <= KS * kQuartet, "the accumulator exchange fits in the staging buffers");
    if (kh == 1) {
#pragma unroll
      for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int p = 0; p < PBW; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) xch[((c * PBW + p) * 16 + r) * 64] = acc[c][p][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int p = 0; p < PBW; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][p][r] += xch[((c * PBW + p) * 16 + r) * 64];
    }
  }
  // ---- epilogue: D[row = cout][col = pixel]; lane: pixel r31 of block p, channels (r&3) + 8*(r>>2) + 4*khalf
  __syncthreads();  // every wave is done with the LDS buffers: the output tile is staged there
  if (KS == 2 && kh == 1) return;
  PXT_STAMP(12);
  const int cw0 = co0 + 32 * CW * wc;
  half_t* out = a.out + (size_t)img * H * W * Cout;
  constexpr int kPitch = CW * 64 + 16;
  char* const stage = smem + wave * (32 * kPitch);
  static_assert(4 * 32 * kPitch <= 2 * kBuf, "the output staging area fits in the halo buffers");
  const int row0 = ty0 + 2 * PBW * wp;
  if (gridDim.z > 1) {
#pragma unroll
    for (int p = 0; p < PBW; ++p) {
      const int gy = row0 + p + (r31 >> 4) * PBW, gx = tx0 + lcol;
      if (!(gy < H && gx < W)) continue;
      float* pd = a.partial + ((((size_t)blockIdx.z * n_img + img) * H + gy) * W + gx) * Cout + cw0;
#pragma unroll
      for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *(float4*)(pd + 32 * c + 8 * g4 + 4 * khalf) =
              make_float4(acc[c][p][4 * g4 + 0], acc[c][p][4 * g4 + 1], acc[c][p][4 * g4 + 2], acc[c][p][4 * g4 + 3]);
    }
    PXT_STAMP(3);
    return;
  }
  if (CW == 1 && a.head.enabled) {  // workgroup-uniform: the fine 1x1 head fused into the last decoder layer (FusedHead)
#pragma unroll
    for (int p = 0; p < PBW; ++p) {
      const int gy = row0 + p + (r31 >> 4) * PBW, gx = tx0 + lcol;
      const bool inside = gy < H && gx < W;
      // bias + ReLU + fp16: the 16 accumulator registers ARE the two B fragments of the head's GEMM (its weight
      // columns were permuted to this register order when the context was created)
      half8 hb[2];
      float cdot = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int chn = (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const half_t hv = (half_t)fmaxf(acc[0][p][r] + a.bias[chn], 0.f);
        hb[r >> 3][r & 7] = hv;
        cdot += (float)hv * a.head.conf_w[chn];
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
        for (int g4 = 0; g4 < 4; ++g4)
          *(float4*)(o + 8 * g4 + 4 * khalf) = make_float4(hd[4 * g4] * inv, hd[4 * g4 + 1] * inv, hd[4 * g4 + 2] * inv, hd[4 * g4 + 3] * inv);
        if (khalf == 0) *(float4*)(o + 32) = make_float4(1.f / (1.f + expf(cdot + a.head.bias[32])), 0.f, 0.f, 0.f);
      }
    }
    PXT_STAMP(3);
    return;
  }
  const bool do_pool = a.pool != nullptr && (PBW % 2 == 0);
  // bias + ReLU + fp16 of block p: 8 * CW packed dwords per lane, [c][g4][2]
  auto pack_block = [&](int p, unsigned (&o)[CW][4][2]) {
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 bv = *(const float4*)(s_bias + 32 * (CW * wc + c) + 8 * g4 + 4 * khalf);
        float v0 = acc[c][p][4 * g4 + 0] + bv.x, v1 = acc[c][p][4 * g4 + 1] + bv.y;
        float v2 = acc[c][p][4 * g4 + 2] + bv.z, v3 = acc[c][p][4 * g4 + 3] + bv.w;
        if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        half4 h;
        h[0] = (half_t)v0; h[1] = (half_t)v1; h[2] = (half_t)v2; h[3] = (half_t)v3;
        o[c][g4][0] = ((const unsigned*)&h)[0];
        o[c][g4][1] = ((const unsigned*)&h)[1];
      }
  };
  // the tile leaves through LDS so that 4 * CW consecutive lanes store one pixel's 64 * CW contiguous bytes
  auto store_block = [&](int p, unsigned (&o)[CW][4][2]) {
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4 += 2) {
        // lanes 0-31 (khalf 0) end up with channels 8 g4 .. 8 g4 + 7, lanes 32-63 with 8 (g4 + 1) .. + 7
        auto s0 = __builtin_amdgcn_permlane32_swap(o[c][g4][0], o[c][g4 + 1][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(o[c][g4][1], o[c][g4 + 1][1], false, false);
        *(uint4*)(stage + r31 * kPitch + (32 * c + 8 * (g4 + khalf)) * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int kLpp = 4 * CW;  // lanes per pixel (16 B each)
#pragma unroll
    for (int it = 0; it < 2 * CW; ++it) {
      const int pix = it * (64 / kLpp) + lane / kLpp, piece = lane % kLpp;
      const uint4 v = *(const uint4*)(stage + pix * kPitch + piece * 16);
      const int oy = row0 + p + (pix >> 4) * PBW, ox = tx0 + (pix & 15);
      if (oy < H && ox < W) *(uint4*)(out + ((size_t)oy * W + ox) * Cout + cw0 + piece * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();  // the next block overwrites the stage
  };
  if (do_pool) {  // rows p and p + 1 (p even) are a pooling pair held by the same lane; the column pair comes by DPP
#pragma unroll
    for (int p = 0; p < PBW - (PBW % 2); p += 2) {
      unsigned o0[CW][4][2], o1[CW][4][2];
      pack_block(p, o0);
      pack_block(p + 1, o1);
      const int gy = row0 + p + (r31 >> 4) * PBW, gx = tx0 + lcol;
      const bool pool_lane = (lcol & 1) == 0 && (gy >> 1) < (H >> 1) && (gx >> 1) < (W >> 1);
      half_t* pdst = a.pool + (((size_t)img * (H >> 1) + (gy >> 1)) * (W >> 1) + (gx >> 1)) * Cout + cw0;
#pragma unroll
      for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          unsigned m[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned v = pk_max(o0[c][g4][h], o1[c][g4][h]);
            const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
            m[h] = pk_max(v, nb);
          }
          if (pool_lane) *(uint2*)(pdst + 32 * c + 8 * g4 + 4 * khalf) = make_uint2(m[0], m[1]);
        }
      store_block(p, o0);
      store_block(p + 1, o1);
    }
  } else {
#pragma unroll
    for (int p = 0; p < PBW; ++p) {
      unsigned o[CW][4][2];
      pack_block(p, o);
      store_block(p, o);
    }
  }
  PXT_STAMP(3);
}

}  // namespace pxt
