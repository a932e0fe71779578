// GroupNorm (channels-last, optional fused SiLU) and LayerNorm.  HBM/L2-bound kernels: every access is a
// full-row coalesced float4 stream; statistics are accumulated in fp32 per thread over a few rows and combined
// in fp64 in a fixed order (deterministic, no atomics).
#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

constexpr int GN_MAX_C = 2560;

__host__ __device__ inline int gn_chunks(int HW) {
  int c = HW / 16;           // >= 16 rows per chunk at 32x32
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  while (HW % c) --c;
  return c;
}

// pass 1: per (batch, row-chunk) partial sum / sum of squares per group.
// grid (chunks, B), 256 threads; thread t owns channels t, t+256, ...
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ ws, int HW, int C,
                                                       int groups, int chunks) {
  __shared__ float s_sum[GN_MAX_C];
  __shared__ float s_sq[GN_MAX_C];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rows = HW / chunks;
  const float* xb = x + ((size_t)b * HW + (size_t)chunk * rows) * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, q = 0.f;
    for (int r = 0; r < rows; ++r) {
      const float v = xb[(size_t)r * C + c];
      s += v;
      q += v * v;
    }
    s_sum[c] = s;
    s_sq[c] = q;
  }
  __syncthreads();
  const int cg = C / groups;
  if (threadIdx.x < groups) {
    double s = 0.0, q = 0.0;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) {
      s += (double)s_sum[c];
      q += (double)s_sq[c];
    }
    double* o = ws + (((size_t)b * chunks + chunk) * groups + threadIdx.x) * 2;
    o[0] = s;
    o[1] = q;
  }
}

// pass 2: finalise mean / rstd per group (fixed-order fp64 sum over chunks), fold gamma/beta into per-channel
// scale/shift (y = x * a_c + b_c, as ATen's CPU GroupNorm does), apply, optional SiLU.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, u16* __restrict__ y_sp,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const double* __restrict__ ws, int HW, int C, int groups, int chunks,
                                                       float eps, int silu) {
  __shared__ float s_a[GN_MAX_C];
  __shared__ float s_b[GN_MAX_C];
  __shared__ float s_mean[64];
  __shared__ float s_rstd[64];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rows = HW / chunks;
  const int cg = C / groups;
  if (threadIdx.x < groups) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < chunks; ++k) {
      const double* o = ws + (((size_t)b * chunks + k) * groups + threadIdx.x) * 2;
      s += o[0];
      q += o[1];
    }
    const double n = (double)HW * cg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[threadIdx.x] = (float)mean;
    s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    const float a = s_rstd[g] * gamma[c];
    s_a[c] = a;
    s_b[c] = beta[c] - s_mean[g] * a;
  }
  __syncthreads();
  const size_t base = ((size_t)b * HW + (size_t)chunk * rows) * C;
  const int n4 = rows * C / 4;
  const float4* x4 = (const float4*)(x + base);
  for (int i = threadIdx.x; i < n4; i += 256) {
    const int c = (i * 4) % C;
    float4 v = x4[i];
    v.x = v.x * s_a[c] + s_b[c];
    v.y = v.y * s_a[c + 1] + s_b[c + 1];
    v.z = v.z * s_a[c + 2] + s_b[c + 2];
    v.w = v.w * s_a[c + 3] + s_b[c + 3];
    if (silu) {
      v.x = silu_f(v.x);
      v.y = silu_f(v.y);
      v.z = silu_f(v.z);
      v.w = silu_f(v.w);
    }
    store_sp4(y_sp, (size_t)b * HW + (size_t)chunk * rows + (size_t)(i * 4) / C, C, c, v.x, v.y, v.z, v.w);
  }
}

// LayerNorm: one wave per row, C <= 1280 (5 float4 per lane); two-pass statistics in registers.
template <int MAXV>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, u16* __restrict__ y_sp,
                                                 const float* __restrict__ w, const float* __restrict__ bia, int rows, int C, float eps,
                                                 int w_plus_one) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int n4 = C >> 2;
  const float4* xr = (const float4*)(x + (size_t)row * C);
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + 64 * i;
    if (idx < n4) {
      v[i] = xr[idx];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + 64 * i;
    if (idx < n4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + 64 * i;
    if (idx < n4) {
      float4 o;
      o.x = (v[i].x - mean) * rstd;
      o.y = (v[i].y - mean) * rstd;
      o.z = (v[i].z - mean) * rstd;
      o.w = (v[i].w - mean) * rstd;
      if (w) {
        float4 ww = ((const float4*)w)[idx];
        if (w_plus_one) {
          ww.x += 1.f;
          ww.y += 1.f;
          ww.z += 1.f;
          ww.w += 1.f;
        }
        o.x *= ww.x;
        o.y *= ww.y;
        o.z *= ww.z;
        o.w *= ww.w;
      }
      if (bia) {
        const float4 bb = ((const float4*)bia)[idx];
        o.x += bb.x;
        o.y += bb.y;
        o.z += bb.z;
        o.w += bb.w;
      }
      store_sp4(y_sp, (size_t)row, C, idx * 4, o.x, o.y, o.z, o.w);
    }
  }
}

}  // namespace

extern "C" int mvd_groupnorm_chunks(int HW) { return gn_chunks(HW); }

extern "C" int mvd_groupnorm_nhwc(const float* x, void* y_sp, const float* gamma, const float* beta, int B, int HW, int C,
                                  int groups, float eps, int silu, double* ws, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && y_sp && gamma && beta && ws, "mvd_groupnorm_nhwc: null pointer");
  MVD_CHECK_ARG(C % 32 == 0, "mvd_groupnorm_nhwc: split-planes output needs C %% 32 == 0 (C=%d)", C);
  MVD_CHECK_ARG(B > 0 && HW > 0 && C > 0 && groups > 0 && groups <= 64 && C % groups == 0, "mvd_groupnorm_nhwc: bad shape");
  MVD_CHECK_ARG(C <= GN_MAX_C && C % 4 == 0, "mvd_groupnorm_nhwc: C=%d must be <= %d and a multiple of 4", C, GN_MAX_C);
  const int chunks = gn_chunks(HW);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, B), dim3(256), 0, s, x, ws, HW, C, groups, chunks);
  MVD_CHECK_LAUNCH("mvd_groupnorm_nhwc/stats");
  hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, B), dim3(256), 0, s, x, (u16*)y_sp, gamma, beta, ws, HW, C, groups, chunks,
                     eps, silu);
  MVD_CHECK_LAUNCH("mvd_groupnorm_nhwc/apply");
  return 0;
}

extern "C" int mvd_layernorm(const float* x, void* y_sp, const float* w, const float* b, int rows, int C, float eps,
                             int w_plus_one, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && y_sp && rows > 0, "mvd_layernorm: bad arguments");
  MVD_CHECK_ARG(C % 32 == 0, "mvd_layernorm: split-planes output needs C %% 32 == 0 (C=%d)", C);
  u16* yh = (u16*)y_sp;
  MVD_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= 1280, "mvd_layernorm: C=%d must be a multiple of 4 and <= 1280", C);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cdiv(rows, 4));
  if (C <= 256)
    hipLaunchKernelGGL(ln_kernel<1>, grid, dim3(256), 0, s, x, yh, w, b, rows, C, eps, w_plus_one);
  else if (C <= 512)
    hipLaunchKernelGGL(ln_kernel<2>, grid, dim3(256), 0, s, x, yh, w, b, rows, C, eps, w_plus_one);
  else
    hipLaunchKernelGGL(ln_kernel<5>, grid, dim3(256), 0, s, x, yh, w, b, rows, C, eps, w_plus_one);
  MVD_CHECK_LAUNCH("mvd_layernorm");
  return 0;
}
