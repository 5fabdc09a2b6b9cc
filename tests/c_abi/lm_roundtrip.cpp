// The drop-in boundary exercised from C++ with nothing but the HIP runtime and
// include/pixtrack_hip.h (no Python, no torch): a caller that owns plain device buffers
//   1. samples sparse reference observations of a synthetic feature map at the true pose
//      (pxt_sample_sparse), so the true pose is the optimum by construction,
//   2. perturbs the pose and refines it back with pxt_lm_refine,
//   3. checks the recovered pose, the output record and the argument-error convention.
// Build (tests/test_c_abi_gpu.py does this): hipcc -Iinclude tests/c_abi/lm_roundtrip.cpp
//   -Lpixtrack_amd -lpixtrack_hip -o lm_roundtrip
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "pixtrack_hip.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);      \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)

static unsigned rng_state = 12345u;
static float frand() {  // uniform [0, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)(rng_state >> 8) * (1.0f / 16777216.0f);
}

static void rodrigues(const float w[3], float R[9]) {
  const float th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const float k[3] = {w[0] / th, w[1] / th, w[2] / th};
  const float c = std::cos(th), s = std::sin(th), v = 1.f - c;
  R[0] = c + k[0] * k[0] * v;        R[1] = k[0] * k[1] * v - k[2] * s; R[2] = k[0] * k[2] * v + k[1] * s;
  R[3] = k[1] * k[0] * v + k[2] * s; R[4] = c + k[1] * k[1] * v;        R[5] = k[1] * k[2] * v - k[0] * s;
  R[6] = k[2] * k[0] * v - k[1] * s; R[7] = k[2] * k[1] * v + k[0] * s; R[8] = c + k[2] * k[2] * v;
}

int main() {
  std::printf("pxt_version %d\n", pxt_version());
  const int H = 120, W = 160, C = 32, CS = 36, N = 1500;
  // smooth C-channel field: sum of a few random plane waves per channel, + confidence 1
  std::vector<float> map((size_t)H * W * CS, 0.f);
  for (int c = 0; c < C; ++c) {
    float fx[3], fy[3], ph[3], am[3];
    for (int k = 0; k < 3; ++k) { fx[k] = (frand() - 0.5f) * 0.25f; fy[k] = (frand() - 0.5f) * 0.25f; ph[k] = frand() * 6.28f; am[k] = 0.5f + frand(); }
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float v = 0.f;
        for (int k = 0; k < 3; ++k) v += am[k] * std::sin(fx[k] * x + fy[k] * y + ph[k]);
        map[((size_t)y * W + x) * CS + c] = v;
      }
  }
  // the query map handed to the LM is L2-normalised per pixel (header: "already normalised")
  std::vector<float> qmap = map;
  for (size_t p = 0; p < (size_t)H * W; ++p) {
    float ss = 0.f;
    for (int c = 0; c < C; ++c) ss += qmap[p * CS + c] * qmap[p * CS + c];
    const float inv = 1.f / std::fmax(std::sqrt(ss), 1e-12f);
    for (int c = 0; c < C; ++c) qmap[p * CS + c] *= inv;
    qmap[p * CS + C] = 1.f;
    map[p * CS + C] = 1.f;
  }
  // points in front of a camera looking down +z from the origin, true pose = small motion
  std::vector<float> p3d((size_t)N * 3);
  for (int i = 0; i < N; ++i) {
    p3d[3 * i + 0] = (frand() - 0.5f) * 1.2f;
    p3d[3 * i + 1] = (frand() - 0.5f) * 0.9f;
    p3d[3 * i + 2] = 2.0f + frand();
  }
  float T_gt[12], T0[12];
  const float w_gt[3] = {0.02f, -0.03f, 0.01f};
  rodrigues(w_gt, T_gt);
  T_gt[9] = 0.01f; T_gt[10] = -0.02f; T_gt[11] = 0.03f;
  const float w0[3] = {0.02f + 0.012f, -0.03f - 0.01f, 0.01f + 0.008f};  // ~1 degree off
  rodrigues(w0, T0);
  T0[9] = 0.01f + 0.01f; T0[10] = -0.02f - 0.008f; T0[11] = 0.03f + 0.012f;
  const float cam[10] = {(float)W, (float)H, 1.2f * W, 1.2f * W, W / 2.f - 0.5f, H / 2.f - 0.5f, 0, 0, 0, 0};

  float *d_map, *d_qmap, *d_p3d, *d_ref, *d_out, *d_log;
  uint8_t* d_valid;
  void* d_ws;
  CK(hipMalloc(&d_map, map.size() * 4));
  CK(hipMalloc(&d_qmap, qmap.size() * 4));
  CK(hipMalloc(&d_p3d, p3d.size() * 4));
  CK(hipMalloc(&d_ref, (size_t)N * CS * 4));
  CK(hipMalloc(&d_valid, N));
  CK(hipMalloc(&d_out, (16 + PXT_MAX_LEVELS) * 4));
  const int num_iters = 50;
  CK(hipMalloc(&d_log, (size_t)num_iters * PXT_LM_LOG_STRIDE * 4));
  CK(hipMalloc(&d_ws, (size_t)pxt_lm_workspace_bytes()));
  CK(hipMemcpy(d_map, map.data(), map.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_qmap, qmap.data(), qmap.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_p3d, p3d.data(), p3d.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0, (16 + PXT_MAX_LEVELS) * 4));
  CK(hipMemset(d_log, 0, (size_t)num_iters * PXT_LM_LOG_STRIDE * 4));
  hipStream_t stream;
  CK(hipStreamCreate(&stream));

  pxt_sample_level sl = {};  // (zero: no window of a larger level)
  sl.fmap = d_map; sl.out = d_ref; sl.h = H; sl.w = W; sl.C = C; sl.cstride = CS; sl.ndist = 0;
  for (int i = 0; i < 10; ++i) sl.cam[i] = cam[i];
  int rc = pxt_sample_sparse(d_p3d, N, T_gt, &sl, 1, 1, 1, d_valid, stream);
  if (rc != PXT_OK) { std::printf("pxt_sample_sparse -> %d (%s)\n", rc, pxt_last_error()); return 3; }

  pxt_lm_level lv;
  lv.fmap = d_qmap; lv.fref = d_ref; lv.h = H; lv.w = W; lv.C = C; lv.cstride = CS; lv.ndist = 0;
  for (int i = 0; i < 10; ++i) lv.cam[i] = cam[i];
  for (int i = 0; i < 6; ++i) lv.lambda[i] = 1e-4f;
  pxt_lm_conf conf;
  conf.num_iters = num_iters; conf.pad = 1; conf.loss = 2; conf.loss_alpha = 0.f; conf.loss_scale = 0.1f;
  conf.grad_stop = 1e-4f; conf.dt_stop = 5e-3f; conf.dR_stop = 5e-2f; conf.min_valid = 10; conf.n_workgroups = 0; conf.spin_limit = 0; conf.path = 0;
  rc = pxt_lm_refine(d_p3d, d_valid, N, &lv, 1, T0, &conf, d_out, d_log, d_ws, stream);
  if (rc != PXT_OK) { std::printf("pxt_lm_refine -> %d (%s)\n", rc, pxt_last_error()); return 4; }
  CK(hipStreamSynchronize(stream));

  float out[16 + PXT_MAX_LEVELS];
  std::vector<uint8_t> valid(N);
  std::vector<float> log((size_t)num_iters * PXT_LM_LOG_STRIDE);
  CK(hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost));
  CK(hipMemcpy(valid.data(), d_valid, N, hipMemcpyDeviceToHost));
  CK(hipMemcpy(log.data(), d_log, log.size() * 4, hipMemcpyDeviceToHost));
  int n_valid = 0;
  for (int i = 0; i < N; ++i) n_valid += valid[i];
  // rotation error: angle of R_out R_gt^T; translation error: |t_out - t_gt|
  float tr = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) tr += out[3 * i + k] * T_gt[3 * i + k];
  const float ang = std::acos(std::fmin(1.f, std::fmax(-1.f, (tr - 1.f) * 0.5f)));
  const float dt = std::sqrt((out[9] - T_gt[9]) * (out[9] - T_gt[9]) + (out[10] - T_gt[10]) * (out[10] - T_gt[10]) +
                             (out[11] - T_gt[11]) * (out[11] - T_gt[11]));
  const int iters = (int)out[16];
  std::printf("valid %d of %d, iterations %d, failed %g, status %g, done %g\n", n_valid, N, iters, out[12], out[13], out[15]);
  std::printf("first cost %.6f last cost %.6f\n", log[0], log[(size_t)(iters - 1) * PXT_LM_LOG_STRIDE]);
  std::printf("rotation error %.3e rad, translation error %.3e\n", ang, dt);
  bool ok = n_valid > N / 2 && out[12] == 0.f && out[13] == 0.f && out[15] == 1.f && iters >= 2 && iters < num_iters &&
            ang < 1e-3f && dt < 1e-3f && log[(size_t)(iters - 1) * PXT_LM_LOG_STRIDE] < log[0];
  // argument-error convention: negative status, no crash
  ok = ok && pxt_lm_refine(nullptr, nullptr, N, &lv, 1, T0, &conf, d_out, nullptr, d_ws, stream) == PXT_E_ARG;
  ok = ok && pxt_lm_refine(d_p3d, nullptr, N, &lv, PXT_MAX_LEVELS + 1, T0, &conf, d_out, nullptr, d_ws, stream) == PXT_E_ARG;
  ok = ok && pxt_sample_sparse(d_p3d, 0, T_gt, &sl, 1, 1, 1, d_valid, stream) == PXT_E_ARG;
  // K problems in one persistent launch (pxt_lm_refine_batch): three copies of the problem, own records and workspaces;
  // every record must equal, bit for bit, a single launch with the same grid
  {
    const int K = 3;
    conf.n_workgroups = 32;
    float one[16 + PXT_MAX_LEVELS];
    CK(hipMemset(d_ws, 0, (size_t)pxt_lm_workspace_bytes()));
    rc = pxt_lm_refine(d_p3d, d_valid, N, &lv, 1, T0, &conf, d_out, nullptr, d_ws, stream);
    CK(hipStreamSynchronize(stream));
    CK(hipMemcpy(one, d_out, sizeof(one), hipMemcpyDeviceToHost));
    ok = ok && rc == PXT_OK && one[15] == 1.f && one[12] == 0.f;
    float* d_outs[K];
    void* d_wss[K];
    void* d_bws = nullptr;
    pxt_lm_problem probs[K];
    for (int k = 0; k < K; ++k) {
      CK(hipMalloc((void**)&d_outs[k], (16 + PXT_MAX_LEVELS) * 4));
      CK(hipMemset(d_outs[k], 0, (16 + PXT_MAX_LEVELS) * 4));
      CK(hipMalloc(&d_wss[k], (size_t)pxt_lm_workspace_bytes()));
      CK(hipMemset(d_wss[k], 0, (size_t)pxt_lm_workspace_bytes()));
      probs[k] = pxt_lm_problem{d_p3d, d_valid, N, &lv, 1, T0, d_outs[k], nullptr, d_wss[k], nullptr};
    }
    CK(hipMalloc(&d_bws, (size_t)pxt_lm_batch_workspace_bytes(K)));
    rc = pxt_lm_refine_batch(probs, K, &conf, d_bws, stream);
    if (rc != PXT_OK) { std::printf("pxt_lm_refine_batch -> %d (%s)\n", rc, pxt_last_error()); return 5; }
    CK(hipStreamSynchronize(stream));
    for (int k = 0; k < K; ++k) {
      float got[16 + PXT_MAX_LEVELS];
      CK(hipMemcpy(got, d_outs[k], sizeof(got), hipMemcpyDeviceToHost));
      ok = ok && std::memcmp(got, one, 17 * sizeof(float)) == 0;
    }
    // two problems on one workspace would read each other's granules: refused
    probs[1].workspace = probs[0].workspace;
    ok = ok && pxt_lm_refine_batch(probs, K, &conf, d_bws, stream) == PXT_E_ARG;
    ok = ok && pxt_lm_refine_batch(probs, PXT_LM_MAX_BATCH + 1, &conf, d_bws, stream) == PXT_E_ARG;
    std::printf("batched launch of %d problems: %s\n", K, ok ? "records equal the single launch" : "MISMATCH");
  }
  std::printf(ok ? "C-ABI ROUNDTRIP OK\n" : "C-ABI ROUNDTRIP FAILED\n");
  return ok ? 0 : 1;
}
