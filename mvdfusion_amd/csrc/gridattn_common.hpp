// Device helpers shared by the GridAttn kernels (gridattn.hip: unfused token kernel; gridattn_fused.hip: fused aggregation):
// camera records, PyTorch3D-convention unproject / project, grid_sample(bilinear, border, align_corners) taps, harmonic embedding.
#pragma once
#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

struct Cam {
  float R[9], T[3], f[2], p[2], C[3];
};

__device__ __forceinline__ Cam load_cam(const float* rec) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.R[i] = rec[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) c.T[i] = rec[9 + i];
  c.f[0] = rec[12];
  c.f[1] = rec[13];
  c.p[0] = rec[14];
  c.p[1] = rec[15];
#pragma unroll
  for (int i = 0; i < 3; ++i) c.C[i] = rec[16 + i];
  return c;
}

// X_world = (X_cam - T) R^T  with X_cam = ((x-px) d / fx, (y-py) d / fy, d)   (pytorch3d unproject_points)
__device__ __forceinline__ void unproject(const Cam& c, float x, float y, float d, float* w) {
  const float xc[3] = {(x - c.p[0]) * d / c.f[0] - c.T[0], (y - c.p[1]) * d / c.f[1] - c.T[1], d - c.T[2]};
#pragma unroll
  for (int j = 0; j < 3; ++j) w[j] = xc[0] * c.R[j * 3 + 0] + xc[1] * c.R[j * 3 + 1] + xc[2] * c.R[j * 3 + 2];
}

// ndc = (fx X/Z + px, fy Y/Z + py) with X_cam = X R + T   (pytorch3d transform_points_ndc)
__device__ __forceinline__ void project(const Cam& c, const float* X, float& u, float& v) {
  float xc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) xc[j] = X[0] * c.R[0 * 3 + j] + X[1] * c.R[1 * 3 + j] + X[2] * c.R[2 * 3 + j] + c.T[j];
  u = c.f[0] * xc[0] / xc[2] + c.p[0];
  v = c.f[1] * xc[1] / xc[2] + c.p[1];
}

// F.grid_sample(bilinear, padding_mode='border', align_corners=True) of 4 consecutive channels at grid (gx, gy)
__device__ __forceinline__ float4 bilinear4(const float* __restrict__ fmap, int S, int ch, float gx, float gy) {
  float ix = ((gx + 1.f) / 2.f) * (float)(S - 1);
  float iy = ((gy + 1.f) / 2.f) * (float)(S - 1);
  ix = fminf(fmaxf(ix, 0.f), (float)(S - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(S - 1));
  if (!(ix == ix)) ix = 0.f;  // NaN coordinates (z ~ 0): stay in bounds
  if (!(iy == iy)) iy = 0.f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - x0f, wy1 = iy - y0f;
  const float wx0 = (x0f + 1.f) - ix, wy0 = (y0f + 1.f) - iy;
  const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t C = 256;
  auto acc = [&](int yy, int xx, float w) {
    if (yy < S && xx < S) {
      const float4 t = *(const float4*)(fmap + ((size_t)yy * S + xx) * C + ch);
      o.x += t.x * w;
      o.y += t.y * w;
      o.z += t.z * w;
      o.w += t.w * w;
    }
  };
  acc(y0, x0, nw);
  acc(y0, x1, ne);
  acc(y1, x0, sw);
  acc(y1, x1, se);
  return o;
}

// harmonic embedding value e of a `dim`-vector: layout [sin(dim*7) | cos(dim*7) | x(dim)], index dim_i*7 + k
__device__ __forceinline__ float harmonic(const float* vec, int dim, int e) {
  const int n = dim * 7;
  if (e >= 2 * n) return vec[e - 2 * n];
  const int ee = e < n ? e : e - n;
  const int di = ee / 7, k = ee - di * 7;
  const float w = 0.1f * (float)(1 << k);  // fl(0.1) * 2^k, as torch computes (2.0**arange(7)) * 0.1
  const float a = vec[di] * w;
  return e < n ? sinf(a) : cosf(a);
}

}  // namespace
