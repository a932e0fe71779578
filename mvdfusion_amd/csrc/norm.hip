// GroupNorm (channels-last, optional fused SiLU) and LayerNorm.  HBM/L2-bound kernels: every access is a
// full-row coalesced float4 stream; statistics are accumulated in fp32 per thread over a few rows and combined
// in fp64 in a fixed order (deterministic, no atomics).
#include <stdlib.h>
#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

constexpr int GN_MAX_C = 2560;

__host__ __device__ inline int gn_chunks(int HW) {
  int c = HW / 8;            // 8 rows per workgroup: >= 1024 workgroups at 32x32 x 8 views
  if (c < 1) c = 1;
  if (c > 128) c = 128;
  while (HW % c) --c;
  return c;
}

// pass 1: per (batch, row-chunk) partial sum / sum of squares per group.
// grid (chunks, B), 256 threads arranged as TY row-slices x TX float4 columns: every global access is a coalesced
// 16-byte load; per-thread fp32 partials over <= rows/TY rows, then a fixed-order fp64 reduction (deterministic).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ ws, int HW, int C,
                                                       int groups, int chunks) {
  __shared__ float s_sum[GN_MAX_C];
  __shared__ float s_sq[GN_MAX_C];
  __shared__ float s_part[2][4][GN_MAX_C / 4];     // used when TY > 1 (C <= 512): [sum|sq][ty][C]
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rows = HW / chunks;
  const int C4 = C >> 2;
  const int TX = C4 < 256 ? C4 : 256;
  int TY = 256 / TX;
  if (TY > 4) TY = 4;
  if (C > GN_MAX_C / 4) TY = 1;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const float4* xb = (const float4*)(x + ((size_t)b * HW + (size_t)chunk * rows) * C);
  if (ty < TY) {
    for (int c4 = tx; c4 < C4; c4 += TX) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = ty; r < rows; r += TY) {
        const float4 v = xb[(size_t)r * C4 + c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
      }
      if (TY == 1) {
        *(float4*)&s_sum[c4 * 4] = s;
        *(float4*)&s_sq[c4 * 4] = q;
      } else {
        *(float4*)&s_part[0][ty][c4 * 4] = s;
        *(float4*)&s_part[1][ty][c4 * 4] = q;
      }
    }
  }
  __syncthreads();
  if (TY > 1) {
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f, q = 0.f;
      for (int t = 0; t < TY; ++t) {
        s += s_part[0][t][c];
        q += s_part[1][t][c];
      }
      s_sum[c] = s;
      s_sq[c] = q;
    }
    __syncthreads();
  }
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
// scale/shift (y = x * a_c + b_c, as ATen's CPU GroupNorm does), apply, optional SiLU, write split planes.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, u16* __restrict__ y_sp,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const double* __restrict__ ws, int HW, int C, int groups, int chunks,
                                                       float eps, int silu) {
  __shared__ float s_a[GN_MAX_C];
  __shared__ float s_b[GN_MAX_C];
  __shared__ double s_ps[8][64], s_pq[8][64];
  __shared__ float s_mean[64];
  __shared__ float s_rstd[64];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rows = HW / chunks;
  const int cg = C / groups;
  {   // 8 slices x groups threads sum the chunk partials (each slice in a fixed order), then a fixed-order combine
    const int g = threadIdx.x % 32, sl = threadIdx.x / 32;
    for (int gg = g; gg < groups; gg += 32) {
      double s = 0.0, q = 0.0;
      for (int k = sl; k < chunks; k += 8) {
        const double* o = ws + (((size_t)b * chunks + k) * groups + gg) * 2;
        s += o[0];
        q += o[1];
      }
      s_ps[sl][gg] = s;
      s_pq[sl][gg] = q;
    }
  }
  __syncthreads();
  if (threadIdx.x < groups) {
    double s = 0.0, q = 0.0;
    for (int sl = 0; sl < 8; ++sl) {
      s += s_ps[sl][threadIdx.x];
      q += s_pq[sl][threadIdx.x];
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
  const size_t row0 = (size_t)b * HW + (size_t)chunk * rows;
  const int C4 = C >> 2;
  const int n4 = rows * C4;
  const float4* x4 = (const float4*)(x + row0 * C);
  for (int i = threadIdx.x; i < n4; i += 256) {
    const int r = i / C4;
    const int c = (i - r * C4) * 4;
    float4 v = x4[i];
    v.x = v.x * s_a[c] + s_b[c];
    v.y = v.y * s_a[c + 1] + s_b[c + 1];
    v.z = v.z * s_a[c + 2] + s_b[c + 2];
    v.w = v.w * s_a[c + 3] + s_b[c + 3];
    if (silu & 2) {   // the VAE decoder tail keeps the GroupNorm output as fp16 (model.py:564-570)
      v.x = (float)(_Float16)v.x;
      v.y = (float)(_Float16)v.y;
      v.z = (float)(_Float16)v.z;
      v.w = (float)(_Float16)v.w;
    }
    if (silu & 1) {
      v.x = silu_f(v.x);
      v.y = silu_f(v.y);
      v.z = silu_f(v.z);
      v.w = silu_f(v.w);
    }
    store_sp4(y_sp, row0 + r, C, c, v.x, v.y, v.z, v.w);
  }
}

// GroupNorm apply with the statistics emitted by the producer of x (gn_stats_add: 2^24 fixed-point {sum, sum of squares} per
// (image, group)): mean / rstd in fp64 exactly as gn_apply_kernel does from its partial sums, then the same apply loop.
__global__ __launch_bounds__(256) void gn_apply_stats_kernel(const float* __restrict__ x, u16* __restrict__ y_sp,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const long long* __restrict__ stats, int HW, int C, int groups, int chunks,
                                                             float eps, int silu) {
  __shared__ float s_a[GN_MAX_C];
  __shared__ float s_b[GN_MAX_C];
  __shared__ float s_mean[64];
  __shared__ float s_rstd[64];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rows = HW / chunks;
  const int cg = C / groups;
  if (threadIdx.x < groups) {
    const long long* p = stats + ((size_t)b * groups + threadIdx.x) * 2;
    const double inv = 1.0 / (double)MVD_GN_FIXED_SCALE;
    const double n = (double)HW * cg;
    const double mean = (double)p[0] * inv / n;
    double var = (double)p[1] * inv / n - mean * mean;
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
  const size_t row0 = (size_t)b * HW + (size_t)chunk * rows;
  const int C4 = C >> 2;
  const int n4 = rows * C4;
  const float4* x4 = (const float4*)(x + row0 * C);
  for (int i = threadIdx.x; i < n4; i += 256) {
    const int r = i / C4;
    const int c = (i - r * C4) * 4;
    float4 v = x4[i];
    v.x = v.x * s_a[c] + s_b[c];
    v.y = v.y * s_a[c + 1] + s_b[c + 1];
    v.z = v.z * s_a[c + 2] + s_b[c + 2];
    v.w = v.w * s_a[c + 3] + s_b[c + 3];
    if (silu & 2) {
      v.x = (float)(_Float16)v.x;
      v.y = (float)(_Float16)v.y;
      v.z = (float)(_Float16)v.z;
      v.w = (float)(_Float16)v.w;
    }
    if (silu & 1) {
      v.x = silu_f(v.x);
      v.y = silu_f(v.y);
      v.z = silu_f(v.z);
      v.w = silu_f(v.w);
    }
    store_sp4(y_sp, row0 + r, C, c, v.x, v.y, v.z, v.w);
  }
}

// The same apply, the default: a thread keeps ONE 16-byte column (TX = the
// largest divisor of C / 4 that fits the block, TY = 256 / TX rows per pass, column slabs in grid.z) and derives the coefficients of its four
// channels itself (fp64 mean / rstd of its one or two groups): no LDS tables, no barriers, no division in the loop, and the tensor spreads over
// 128 - 512 blocks instead of the 16 - 64 of the row-chunk kernel, whose per-block prologue (tables for ALL channels behind two barriers) took
// longer than the data: 8 x 4x4 x 2560: 10.5 -> 3.4 us, 8 x 8x8 x 2560: 11.0 -> 4.6, 8 x 16x16 x 1920: 10.9 -> 7.7, 8 x 32x32 x 960: 13.8 -> 13.6; only
// 320-channel tensors of 32 x 32 and larger images stay on the row-chunk kernel (6.3 vs 6.8 us) -- profiles/r03_gn_apply_time.log.  Same
// arithmetic per element => identical results.
__global__ __launch_bounds__(256) void gn_apply_stats_cols_kernel(const float* __restrict__ x, u16* __restrict__ y_sp,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const long long* __restrict__ stats, int HW, int C, int groups, int rows_per_block,
                                                             int TX, float eps, int silu) {
  const int b = blockIdx.y;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  if (ty >= TY) return;
  const int C4 = C >> 2, cg = C / groups;
  const int col4 = blockIdx.z * TX + tx, c = col4 * 4;
  const int r_end = min(HW, ((int)blockIdx.x + 1) * rows_per_block);
  int r = blockIdx.x * rows_per_block + ty;
  const float4* xp = (const float4*)x + ((size_t)b * HW + r) * C4 + col4;
  const size_t xstep = (size_t)TY * C4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < r_end) v = *xp;                                  // in flight under the coefficient arithmetic below
  const float4 gm = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
  float a[4], bb[4];
  const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
  int gprev = -1;
  float mk = 0.f, rk = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = (c + k) / cg;
    if (g != gprev) {                                      // (a column's four channels lie in one or two groups unless the groups are narrower than 4)
      const long long* p = stats + ((size_t)b * groups + g) * 2;
      const double inv = 1.0 / (double)MVD_GN_FIXED_SCALE;
      const double n = (double)HW * cg;
      const double m = (double)p[0] * inv / n;
      double var = (double)p[1] * inv / n - m * m;
      if (var < 0.0) var = 0.0;
      mk = (float)m;
      rk = (float)(1.0 / sqrt(var + (double)eps));
      gprev = g;
    }
    a[k] = rk * gmv[k];
    bb[k] = btv[k] - mk * a[k];
  }
  for (; r < r_end; r += TY) {
    float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r + TY < r_end) vn = *(xp + xstep);
    xp += xstep;
    v.x = v.x * a[0] + bb[0];
    v.y = v.y * a[1] + bb[1];
    v.z = v.z * a[2] + bb[2];
    v.w = v.w * a[3] + bb[3];
    if (silu & 2) {
      v.x = (float)(_Float16)v.x;
      v.y = (float)(_Float16)v.y;
      v.z = (float)(_Float16)v.z;
      v.w = (float)(_Float16)v.w;
    }
    if (silu & 1) {
      v.x = silu_f(v.x);
      v.y = silu_f(v.y);
      v.z = silu_f(v.z);
      v.w = silu_f(v.w);
    }
    store_sp4(y_sp, (size_t)b * HW + r, C, c, v.x, v.y, v.z, v.w);
    v = vn;
  }
}

// Row softmax of a (rows, cols) fp32 logit matrix, scaled first; probabilities written as split planes (the A operand of
// the P*V GEMM).  One wave per row, cols <= 4096 held in registers (the VAE AttnBlock has one head over h*w <= 4096 keys).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, u16* __restrict__ y_sp, int rows, int cols,
                                                           int ldx, float scale, float out_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + (size_t)row * ldx);
  const int n4 = cols >> 2;
  float4 v[16];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = lane + i * 64;
    if (idx < n4) {
      v[i] = xr[idx];
      v[i].x *= scale; v[i].y *= scale; v[i].z *= scale; v[i].w *= scale;
      mx = fmaxf(fmaxf(fmaxf(mx, v[i].x), fmaxf(v[i].y, v[i].z)), v[i].w);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = lane + i * 64;
    if (idx < n4) {
      v[i].x = expf(v[i].x - mx); v[i].y = expf(v[i].y - mx); v[i].z = expf(v[i].z - mx); v[i].w = expf(v[i].w - mx);
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  sum = wave_sum(sum);
  const float inv = out_scale / sum;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = lane + i * 64;
    if (idx < n4) store_sp4(y_sp, (size_t)row, cols, idx * 4, v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
  }
}

// LayerNorm: one wave per row, C <= 1280 (5 float4 per lane); two-pass statistics in registers.
template <int MAXV>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, u16* __restrict__ y_sp, float* __restrict__ y_f32,
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
      if (y_sp) store_sp4(y_sp, (size_t)row, C, idx * 4, o.x, o.y, o.z, o.w);
      if (y_f32) ((float4*)(y_f32 + (size_t)row * C))[idx] = o;
    }
  }
}

}  // namespace

extern "C" int mvd_groupnorm_chunks(int HW) { return gn_chunks(HW); }

extern "C" int mvd_groupnorm_nhwc(const float* x, void* y_sp, const float* gamma, const float* beta, int B, int HW, int C,
                                  int groups, float eps, int silu, double* ws, size_t ws_elems, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && y_sp && gamma && beta && ws, "mvd_groupnorm_nhwc: null pointer");
  MVD_CHECK_ARG(C % 32 == 0, "mvd_groupnorm_nhwc: split-planes output needs C %% 32 == 0 (C=%d)", C);
  MVD_CHECK_ARG(B > 0 && HW > 0 && C > 0 && groups > 0 && groups <= 64 && C % groups == 0, "mvd_groupnorm_nhwc: bad shape");
  MVD_CHECK_ARG(C <= GN_MAX_C && C % 4 == 0, "mvd_groupnorm_nhwc: C=%d must be <= %d and a multiple of 4", C, GN_MAX_C);
  const int chunks = gn_chunks(HW);
  MVD_CHECK_ARG((size_t)B * chunks * groups * 2 <= ws_elems,
                "mvd_groupnorm_nhwc: workspace holds %zu doubles, B*chunks*groups*2 = %zu needed (B=%d, chunks=%d, groups=%d)", ws_elems,
                (size_t)B * chunks * groups * 2, B, chunks, groups);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, B), dim3(256), 0, s, x, ws, HW, C, groups, chunks);
  MVD_CHECK_LAUNCH("mvd_groupnorm_nhwc/stats");
  hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, B), dim3(256), 0, s, x, (u16*)y_sp, gamma, beta, ws, HW, C, groups, chunks,
                     eps, silu);
  MVD_CHECK_LAUNCH("mvd_groupnorm_nhwc/apply");
  return 0;
}

extern "C" int mvd_groupnorm_from_stats(const float* x, void* y_sp, const float* gamma, const float* beta, const long long* stats, int B,
                                        int HW, int C, int groups, float eps, int silu, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && y_sp && gamma && beta && stats, "mvd_groupnorm_from_stats: null pointer");
  MVD_CHECK_ARG(C % 32 == 0 && B > 0 && HW > 0 && groups > 0 && groups <= 64 && C % groups == 0 && C <= GN_MAX_C,
                "mvd_groupnorm_from_stats: bad shape (C=%d groups=%d)", C, groups);
  // the column-per-thread kernel everywhere except the widest-and-thinnest tensors (32 x 32 and larger images with <= 320 channels: 80
  // columns leave 16 of 256 threads idle and the row-chunk kernel is 3 - 8 % faster there; profiles/r03_gn_apply_time.log)
  if (!(C <= 320 && HW >= 1024)) {
    const int C4 = C / 4;
    int TX = C4 < 256 ? C4 : 256;
    while (C4 % TX) --TX;                                  // largest divisor of C / 4 that fits a 256-thread block
    const int TY = 256 / TX;
    const int rows_per_block = TY * 4;                     // <= 4 rows per thread
    hipLaunchKernelGGL(gn_apply_stats_cols_kernel, dim3(cdiv(HW, rows_per_block), B, C4 / TX), dim3(256), 0, (hipStream_t)stream, x,
                       (u16*)y_sp, gamma, beta, stats, HW, C, groups, rows_per_block, TX, eps, silu);
  } else {
    const int chunks = gn_chunks(HW);
    hipLaunchKernelGGL(gn_apply_stats_kernel, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, x, (u16*)y_sp, gamma, beta, stats, HW, C,
                       groups, chunks, eps, silu);
  }
  MVD_CHECK_LAUNCH("mvd_groupnorm_from_stats");
  return 0;
}

extern "C" int mvd_softmax_rows(const float* x, void* y_sp, int rows, int cols, int ldx, float scale, float out_scale,
                                mvd_stream_t stream) {
  MVD_CHECK_ARG(x && y_sp && rows > 0 && cols > 0, "mvd_softmax_rows: bad arguments");
  MVD_CHECK_ARG(cols % 32 == 0 && cols <= 4096 && ldx % 4 == 0 && ldx >= cols, "mvd_softmax_rows: cols=%d must be a multiple of 32, <= 4096", cols);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y_sp, rows, cols, ldx, scale,
                     out_scale);
  MVD_CHECK_LAUNCH("mvd_softmax_rows");
  return 0;
}

extern "C" int mvd_layernorm(const float* x, void* y_sp, float* y_f32, const float* w, const float* b, int rows, int C, float eps,
                             int w_plus_one, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && (y_sp || y_f32) && rows > 0, "mvd_layernorm: bad arguments");
  MVD_CHECK_ARG(C % 32 == 0, "mvd_layernorm: split-planes output needs C %% 32 == 0 (C=%d)", C);
  u16* yh = (u16*)y_sp;
  MVD_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= 1280, "mvd_layernorm: C=%d must be a multiple of 4 and <= 1280", C);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cdiv(rows, 4));
  if (C <= 256)
    hipLaunchKernelGGL(ln_kernel<1>, grid, dim3(256), 0, s, x, yh, y_f32, w, b, rows, C, eps, w_plus_one);
  else if (C <= 512)
    hipLaunchKernelGGL(ln_kernel<2>, grid, dim3(256), 0, s, x, yh, y_f32, w, b, rows, C, eps, w_plus_one);
  else
    hipLaunchKernelGGL(ln_kernel<5>, grid, dim3(256), 0, s, x, yh, y_f32, w, b, rows, C, eps, w_plus_one);
  MVD_CHECK_LAUNCH("mvd_layernorm");
  return 0;
}
