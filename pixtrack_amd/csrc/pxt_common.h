// Shared host/device helpers for the gfx950 kernels of the pixtrack hot path.
// HIP/CDNA4 only: 64-lane wavefronts are assumed throughout (no other target).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pixtrack_hip.h"

#define PXT_WAVE 64

namespace pxt {

// ---- error plumbing --------------------------------------------------------
void set_last_error(const char* what, hipError_t e);
hipStream_t shared_side_stream(int i);  // i in 0..2, per device, shared by all contexts (pxt_core.hip)
#define PXT_HIP_CHECK(expr)                         \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) {                         \
      ::pxt::set_last_error(#expr, _e);             \
      return PXT_E_HIP;                             \
    }                                               \
  } while (0)

// ---- camera model (pixloc Camera.world2image / J_world2image) ---
struct Cam {
  float w, h, fx, fy, cx, cy, k1, k2, p1, p2;
  int ndist;
};

__host__ __device__ inline Cam make_cam(const float* c10, int ndist) {
  Cam c;
  c.w = c10[0]; c.h = c10[1]; c.fx = c10[2]; c.fy = c10[3]; c.cx = c10[4]; c.cy = c10[5];
  c.k1 = c10[6]; c.k2 = c10[7]; c.p1 = c10[8]; c.p2 = c10[9];
  c.ndist = ndist;
  return c;
}

constexpr float kCamEps = 1e-3f;

// Projects a camera-frame point.  Returns validity (z > eps, inside the
// distortion model's monotone range, inside the image); writes pixel coords and,
// if J != nullptr, the 2x3 Jacobian d(u,v)/d(p) in row-major order.
__device__ inline bool project_point(const Cam& c, float x, float y, float z, float& u, float& v,
                                     float* J /* 6 or nullptr */) {
  bool visible = z > kCamEps;
  float zc = fmaxf(z, kCamEps);
  float iz = 1.0f / zc;
  float xn = x * iz, yn = y * iz;
  float xd = xn, yd = yn;
  bool dist_ok = true;
  float Jd00 = 1.f, Jd01 = 0.f, Jd10 = 0.f, Jd11 = 1.f;
  if (c.ndist > 0) {
    float r2 = xn * xn + yn * yn;
    float radial = c.k1 * r2 + c.k2 * r2 * r2;
    xd = xn + xn * radial;
    yd = yn + yn * radial;
    float disc = 9.f * c.k1 * c.k1 - 20.f * c.k2;
    bool limited = ((c.k2 > 0.f) && (disc > 0.f)) || ((c.k2 <= 0.f) && (c.k1 > 0.f));
    if (limited) {
      float limit = (c.k2 > 0.f) ? (sqrtf(fmaxf(disc, 0.f)) - 3.f * c.k1) / (10.f * c.k2)
                                 : 1.f / (3.f * c.k1);
      dist_ok = r2 < fabsf(limit);
    }
    float uv = xn * yn;
    float d_radial = 2.f * c.k1 + 4.f * c.k2 * r2;
    Jd00 += radial + xn * xn * d_radial;
    Jd11 += radial + yn * yn * d_radial;
    Jd01 += uv * d_radial;
    Jd10 += uv * d_radial;
    if (c.ndist > 2) {
      xd += 2.f * c.p1 * uv + c.p2 * (r2 + 2.f * xn * xn);
      yd += 2.f * c.p2 * uv + c.p1 * (r2 + 2.f * yn * yn);
      Jd00 += 2.f * c.p1 * yn + 6.f * c.p2 * xn;
      Jd11 += 2.f * c.p2 * xn + 6.f * c.p1 * yn;
      Jd01 += 2.f * c.p1 * xn + 2.f * c.p2 * yn;
      Jd10 += 2.f * c.p2 * yn + 2.f * c.p1 * xn;
    }
  }
  u = xd * c.fx + c.cx;
  v = yd * c.fy + c.cy;
  bool in_img = (u >= 0.f) && (v >= 0.f) && (u <= c.w - 1.f) && (v <= c.h - 1.f);
  if (J) {
    // J_project rows: [1/z, 0, -x/z^2], [0, 1/z, -y/z^2] with the clamped z.
    float a = iz, bx = -x * iz * iz, by = -y * iz * iz;
    // (diag(f) * Jd) * Jproj
    float m00 = c.fx * Jd00, m01 = c.fx * Jd01, m10 = c.fy * Jd10, m11 = c.fy * Jd11;
    J[0] = m00 * a; J[1] = m01 * a; J[2] = m00 * bx + m01 * by;
    J[3] = m10 * a; J[4] = m11 * a; J[5] = m10 * bx + m11 * by;
  }
  return visible && dist_ok && in_img;
}

// Wave-level butterfly sum over `width` consecutive lanes (width a power of two).
template <int WIDTH>
__device__ inline float group_allreduce_sum(float v) {
#pragma unroll
  for (int m = 1; m < WIDTH; m <<= 1) v += __shfl_xor(v, m, PXT_WAVE);
  return v;
}


// ---- pose -> renderer camera -------------------------------------------------------------------------------
// Pose (world -> camera, row-major R then t: the LM kernel's record) -> the NeRF renderer's camera (3x4, ngp
// coordinates), in the float64 arithmetic of the host chain it stands in for: get_camera_in_world_from_pixpose
// (pose_utils.py:24), sfm_to_nerf_pose (ingp_utils.py:47-63), instant-ngp's nerf_matrix_to_ngp.  The host recomputes
// the camera when the pose reaches it and compares: a render that ran ahead of the host is used only if the 12 floats
// are the same bits - hence no FMA contraction in here, whatever the file's flags (numpy does not fuse).
struct PoseConv {
  double centroid[3], scale3_over_avglen, Rn[16], totp[3], ngp_scale, ngp_offset[3];
};
__host__ inline PoseConv make_pose_conv(const double* conv27) {
  PoseConv cv;
  for (int i = 0; i < 3; ++i) cv.centroid[i] = conv27[i];
  cv.scale3_over_avglen = conv27[3];
  for (int i = 0; i < 16; ++i) cv.Rn[i] = conv27[4 + i];
  for (int i = 0; i < 3; ++i) cv.totp[i] = conv27[20 + i];
  cv.ngp_scale = conv27[23];
  for (int i = 0; i < 3; ++i) cv.ngp_offset[i] = conv27[24 + i];
  return cv;
}
__device__ inline void pose_to_camera_f64(const float* pose12, const PoseConv& cv, float* cam12) {
#pragma clang fp contract(off)
  double R[9], t[3];
  for (int i = 0; i < 9; ++i) R[i] = (double)pose12[i];
  for (int i = 0; i < 3; ++i) t[i] = (double)pose12[9 + i];
  // camera in world: [R^T | (-R^T) t]
  double c[16];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c[4 * i + j] = R[3 * j + i];
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) acc += (-R[3 * k + i]) * t[k];
    c[4 * i + 3] = acc;
  }
  c[12] = c[13] = c[14] = 0.0; c[15] = 1.0;
  // sfm_to_nerf_pose: camera y/z flip (columns 1, 2), rows 0 <-> 1, row 2 negated, recentre, scale, rotate, recentre
  for (int i = 0; i < 4; ++i) { c[4 * i + 1] = -c[4 * i + 1]; c[4 * i + 2] = -c[4 * i + 2]; }
  for (int j = 0; j < 4; ++j) { const double a = c[j]; c[j] = c[4 + j]; c[4 + j] = a; }
  for (int j = 0; j < 4; ++j) c[8 + j] = -c[8 + j];
  for (int i = 0; i < 3; ++i) { c[4 * i + 3] -= cv.centroid[i]; c[4 * i + 3] *= cv.scale3_over_avglen; }
  double p[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += cv.Rn[4 * i + k] * c[4 * k + j];
      p[4 * i + j] = acc;
    }
  for (int i = 0; i < 3; ++i) p[4 * i + 3] -= cv.totp[i];
  // nerf_matrix_to_ngp: flip camera y/z, scale + offset the origin, rows (x, y, z) <- (y, z, x)
  double m[12];
  for (int i = 0; i < 3; ++i) {
    m[4 * i + 0] = p[4 * i + 0];
    m[4 * i + 1] = -p[4 * i + 1];
    m[4 * i + 2] = -p[4 * i + 2];
    m[4 * i + 3] = p[4 * i + 3] * cv.ngp_scale + cv.ngp_offset[i];
  }
  const int perm[3] = {1, 2, 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) cam12[4 * i + j] = (float)m[4 * perm[i] + j];
}

}  // namespace pxt
