// Backward kernels of the conv / linear / GroupNorm family (the training step; reference
// train.py:90-95 `loss.backward()` over mvdfusion/unet.py + openaimodel.py ResBlock / the output head).
//
// The matrix products of a backward pass run on the forward's split-operand MFMA GEMM (gemm.hip):
//   dgrad   dX = dY W         -> mvd_gemm over the planes of dY and the packed TRANSPOSED weight (3x3: the 180-degree rotated,
//                                channel-swapped filter through the same implicit-GEMM address generator)
//   wgrad   dW = dY^T X       -> mvd_gemm with both operands activations (MVD_B_PLANES): A = (dY)^T, B = (X)^T / (im2col X)^T.
// The GEMM reads operands row-major along its reduction dimension, which for wgrad is the ROW index of dY and X -- the kernels
// here produce those transposed split planes (and the im2col'd transposed planes of a 3x3 conv input, rows ordered ci * 9 + tap
// so that the product lands in nn.Conv2d's (Cout, Cin, 3, 3) memory layout), the bias gradient (column sums) and the
// GroupNorm(+SiLU) backward.  Everything is deterministic (fixed-order reductions, no floating-point atomics).
#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

// (rows, cols) -> split planes of the transpose: out row c, k-block rb holds rows 32 rb .. 32 rb + 31 of column c.
// src_planes == 0: x is fp32 with leading dimension ldx; 1: x is split planes (rows, 2 * ldx).
__global__ __launch_bounds__(256) void transpose_planes_kernel(const void* __restrict__ x, int src_planes, int rows, int cols, int ldx,
                                                               u16* __restrict__ out, int ldo) {
  __shared__ unsigned int tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    u16 hi = 0, lo = 0;
    if (r < rows && c < cols) {
      if (src_planes) {
        const u16* p = (const u16*)x + (size_t)r * 2 * ldx + (c >> 5) * 64 + (c & 31);
        hi = p[0];
        lo = p[32];
      } else {
        split_op16(((const float*)x)[(size_t)r * ldx + c], hi, lo);
      }
    }
    tile[ty + 8 * i][tx] = (unsigned int)hi | ((unsigned int)lo << 16);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;        // output row
    if (c >= cols) continue;
    const unsigned int v = tile[tx][ty + 8 * i];
    u16* o = out + (size_t)c * 2 * ldo + (size_t)blockIdx.x * 64 + tx;
    o[0] = (u16)(v & 0xffffu);
    o[32] = (u16)(v >> 16);
  }
}

// 3x3 / stride 1 / pad 1 im2col of a channels-last activation in split planes (B*H*W, 2*Cin), transposed: out row ci * 9 + tap,
// column m (the output pixel); the tap's source pixel outside the image contributes zero.
__global__ __launch_bounds__(256) void im2col_t_kernel(const u16* __restrict__ x, int B, int H, int W, int Cin, u16* __restrict__ out,
                                                       int ldo) {
  __shared__ unsigned int tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tap = blockIdx.z;
  const int ky = tap / 3, kx = tap - ky * 3;
  const int M = B * H * W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 8 * i, c = c0 + tx;
    unsigned int v = 0;
    if (m < M && c < Cin) {
      const int b = m / (H * W), rem = m - b * H * W;
      const int oy = rem / W, ox = rem - oy * W;
      const int iy = oy + ky - 1, ix = ox + kx - 1;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const u16* p = x + ((size_t)(b * H + iy) * W + ix) * 2 * Cin + (c >> 5) * 64 + (c & 31);
        v = (unsigned int)p[0] | ((unsigned int)p[32] << 16);
      }
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    if (c >= Cin) continue;
    const unsigned int v = tile[tx][ty + 8 * i];
    u16* o = out + ((size_t)c * 9 + tap) * 2 * ldo + (size_t)blockIdx.x * 64 + tx;
    o[0] = (u16)(v & 0xffffu);
    o[32] = (u16)(v >> 16);
  }
}

// column sums (bias gradient): grid (cols / 32, splits); thread = (8 row lanes, 32 columns), fp64 partials, fixed-order combine
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, int rows, int cols, int ldx,
                                                              double* __restrict__ part) {
  __shared__ double s[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double a = 0.0;
  if (c < cols)
    for (int r = r0 + ty; r < r1; r += 8) a += (double)x[(size_t)r * ldx + c];
  s[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && c < cols) {
    double t = 0.0;
    for (int k = 0; k < 8; ++k) t += s[k][tx];
    part[(size_t)blockIdx.y * cols + c] = t;
  }
}
__global__ __launch_bounds__(256) void col_sum_final_kernel(const double* __restrict__ part, int splits, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  double t = 0.0;
  for (int k = 0; k < splits; ++k) t += part[(size_t)k * cols + c];
  out[c] = (float)t;
}

// ---------------------------------------------------------------------------------------------- GroupNorm (+ SiLU) backward
// y = act(xhat * gamma + beta), xhat = (x - mean_g) * rstd_g per (image, group);  act = SiLU or identity.
__device__ __forceinline__ float silu_grad(float z) {
  const float sg = 1.0f / (1.0f + __expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

// moments per (image, group): grid (groups, B); fixed-order fp64 tree
__global__ __launch_bounds__(256) void gn_moments_kernel(const float* __restrict__ x, int HW, int C, int groups, float eps,
                                                         float* __restrict__ mom) {
  __shared__ double ss[256], sq[256];
  const int g = blockIdx.x, b = blockIdx.y, cg = C / groups;
  const size_t n = (size_t)HW * cg;
  double s = 0.0, q = 0.0;
  for (size_t e = threadIdx.x; e < n; e += 256) {
    const size_t r = e / cg;
    const int c = g * cg + (int)(e - r * cg);
    const double v = (double)x[((size_t)b * HW + r) * C + c];
    s += v;
    q += v * v;
  }
  ss[threadIdx.x] = s;
  sq[threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      ss[threadIdx.x] += ss[threadIdx.x + o];
      sq[threadIdx.x] += sq[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = ss[0] / (double)n;
    double var = sq[0] / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    mom[((size_t)b * groups + g) * 2] = (float)mean;
    mom[((size_t)b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// per (image, channel): s1 = sum_rows dz, s2 = sum_rows dz * xhat  (dz = dy * act'(z)).  grid (C / 32, B).
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mom, int HW, int C, int groups, int silu,
                                                            float* __restrict__ sums) {
  __shared__ double a1[8][32], a2[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx, b = blockIdx.y, cg = C / groups;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const int g = c / cg;
    const float mean = mom[((size_t)b * groups + g) * 2], rstd = mom[((size_t)b * groups + g) * 2 + 1];
    const float ga = gamma[c], be = beta[c];
    for (int r = ty; r < HW; r += 8) {
      const size_t i = ((size_t)b * HW + r) * C + c;
      const float xh = (x[i] - mean) * rstd;
      float dz = dy[i];
      if (silu) dz *= silu_grad(xh * ga + be);
      s1 += (double)dz;
      s2 += (double)dz * (double)xh;
    }
  }
  a1[ty][tx] = s1;
  a2[ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && c < C) {
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < 8; ++k) {
      t1 += a1[k][tx];
      t2 += a2[k][tx];
    }
    sums[((size_t)b * C + c) * 2] = (float)t1;
    sums[((size_t)b * C + c) * 2 + 1] = (float)t2;
  }
}

// dx = rstd * (gamma * dz - (xhat * ds + db) / n), ds = sum_{c in g} gamma_c s2_c, db = sum_{c in g} gamma_c s1_c.  grid (chunks, B).
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mom, const float* __restrict__ sums, int HW, int C,
                                                           int groups, int chunks, int silu, float* __restrict__ dx) {
  __shared__ float s_ds[64], s_db[64];
  const int chunk = blockIdx.x, b = blockIdx.y, cg = C / groups;
  if (threadIdx.x < groups) {
    double ds = 0.0, db = 0.0;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) {
      ds += (double)gamma[c] * (double)sums[((size_t)b * C + c) * 2 + 1];
      db += (double)gamma[c] * (double)sums[((size_t)b * C + c) * 2];
    }
    s_ds[threadIdx.x] = (float)ds;
    s_db[threadIdx.x] = (float)db;
  }
  __syncthreads();
  const int rows = HW / chunks;
  const size_t base = ((size_t)b * HW + (size_t)chunk * rows) * C;
  const float inv_n = 1.0f / ((float)HW * (float)cg);
  for (int i = threadIdx.x; i < rows * C; i += 256) {
    const int c = i % C, g = c / cg;
    const float mean = mom[((size_t)b * groups + g) * 2], rstd = mom[((size_t)b * groups + g) * 2 + 1];
    const float xh = (x[base + i] - mean) * rstd;
    float dz = dy[base + i];
    if (silu) dz *= silu_grad(xh * gamma[c] + beta[c]);
    dx[base + i] = rstd * (gamma[c] * dz - (xh * s_ds[g] + s_db[g]) * inv_n);
  }
}

// dgamma_c = sum_b s2[b][c], dbeta_c = sum_b s1[b][c]
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ sums, int B, int C, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double g = 0.0, be = 0.0;
  for (int b = 0; b < B; ++b) {
    be += (double)sums[((size_t)b * C + c) * 2];
    g += (double)sums[((size_t)b * C + c) * 2 + 1];
  }
  dgamma[c] = (float)g;
  dbeta[c] = (float)be;
}

}  // namespace

extern "C" int mvd_transpose_planes(const void* x, int src_planes, int rows, int cols, int ldx, void* out_sp, int ldo,
                                    mvd_stream_t stream) {
  MVD_CHECK_ARG(x && out_sp && rows > 0 && cols > 0 && ldx >= cols, "mvd_transpose_planes: bad arguments");
  MVD_CHECK_ARG(ldo % 32 == 0 && ldo >= (rows + 31) / 32 * 32 && ((uintptr_t)out_sp & 127) == 0,
                "mvd_transpose_planes: ldo=%d must be a multiple of 32 covering %d rows, 128-byte aligned output", ldo, rows);
  if (src_planes) MVD_CHECK_ARG(ldx % 32 == 0, "mvd_transpose_planes: plane source needs ldx %% 32 == 0");
  hipLaunchKernelGGL(transpose_planes_kernel, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, src_planes,
                     rows, cols, ldx, (u16*)out_sp, ldo);
  MVD_CHECK_LAUNCH("mvd_transpose_planes");
  return 0;
}

extern "C" int mvd_im2col3x3_t_planes(const void* x_sp, int B, int H, int W, int Cin, void* out_sp, int ldo, mvd_stream_t stream) {
  MVD_CHECK_ARG(x_sp && out_sp && B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 32 == 0, "mvd_im2col3x3_t_planes: bad arguments");
  const int M = B * H * W;
  MVD_CHECK_ARG(ldo % 32 == 0 && ldo >= (M + 31) / 32 * 32 && ((uintptr_t)out_sp & 127) == 0,
                "mvd_im2col3x3_t_planes: ldo=%d must be a multiple of 32 covering %d pixels", ldo, M);
  hipLaunchKernelGGL(im2col_t_kernel, dim3((M + 31) / 32, Cin / 32, 9), dim3(256), 0, (hipStream_t)stream, (const u16*)x_sp, B, H, W, Cin,
                     (u16*)out_sp, ldo);
  MVD_CHECK_LAUNCH("mvd_im2col3x3_t_planes");
  return 0;
}

extern "C" size_t mvd_col_sum_workspace_doubles(int rows, int cols) {
  int splits = rows / 256;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  return (size_t)splits * cols;
}

extern "C" int mvd_col_sum(const float* x, int rows, int cols, int ldx, float* out, double* ws, size_t ws_doubles, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && out && ws && rows > 0 && cols > 0 && ldx >= cols, "mvd_col_sum: bad arguments");
  int splits = rows / 256;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  MVD_CHECK_ARG(ws_doubles >= (size_t)splits * cols, "mvd_col_sum: workspace too small (%zu < %zu doubles)", ws_doubles,
                (size_t)splits * cols);
  hipLaunchKernelGGL(col_sum_partial_kernel, dim3((cols + 31) / 32, splits), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ldx, ws);
  hipLaunchKernelGGL(col_sum_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, splits, cols, out);
  MVD_CHECK_LAUNCH("mvd_col_sum");
  return 0;
}

extern "C" int mvd_groupnorm_backward(const float* x, const float* dy, const float* gamma, const float* beta, int B, int HW, int C,
                                      int groups, float eps, int silu, float* dx, float* dgamma, float* dbeta, float* ws,
                                      size_t ws_floats, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && dy && gamma && beta && dx && dgamma && dbeta && ws, "mvd_groupnorm_backward: null pointer");
  MVD_CHECK_ARG(B > 0 && HW > 0 && C > 0 && groups > 0 && groups <= 64 && C % groups == 0, "mvd_groupnorm_backward: bad shape");
  const size_t need = (size_t)B * groups * 2 + (size_t)B * C * 2;
  MVD_CHECK_ARG(ws_floats >= need, "mvd_groupnorm_backward: workspace too small (%zu < %zu floats)", ws_floats, need);
  float* mom = ws;
  float* sums = ws + (size_t)B * groups * 2;
  hipStream_t s = (hipStream_t)stream;
  int chunks = HW / 8;
  if (chunks < 1) chunks = 1;
  if (chunks > 128) chunks = 128;
  while (HW % chunks) --chunks;
  hipLaunchKernelGGL(gn_moments_kernel, dim3(groups, B), dim3(256), 0, s, x, HW, C, groups, eps, mom);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((C + 31) / 32, B), dim3(256), 0, s, x, dy, gamma, beta, mom, HW, C, groups, silu, sums);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(chunks, B), dim3(256), 0, s, x, dy, gamma, beta, mom, sums, HW, C, groups, chunks, silu, dx);
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, B, C, dgamma, dbeta);
  MVD_CHECK_LAUNCH("mvd_groupnorm_backward");
  return 0;
}
