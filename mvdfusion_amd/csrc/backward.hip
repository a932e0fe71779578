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
                                                               u16* __restrict__ out, int ldo, const float* __restrict__ scale) {
  __shared__ unsigned int tile[32][33];
  const float sc = scale != nullptr ? *scale : 1.f;
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
        split_op16(((const float*)x)[(size_t)r * ldx + c] * sc, hi, lo);
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

// column sums (bias gradient): grid (cols / 32, splits); thread = (8 row lanes, 32 columns), fp64 partials, fixed-order combine.
// amax != null (round 6, mvd_col_sum_pow2): the same pass also forms max|x| -- block maxima meet in an atomicMax on the bit pattern (non-negative
// floats order like their bits) -- and the FINAL kernel turns it into the power-of-two gradient scale {s, 1/s} of mvd_pow2_scale and re-zeroes
// the word: the bias gradient and the operand scale of a layer's dY cost one pass over dY instead of two.
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, int rows, int cols, int ldx,
                                                              double* __restrict__ part, unsigned* __restrict__ amax) {
  __shared__ double s[8][32];
  __shared__ unsigned s_m[4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double a = 0.0;
  unsigned m = 0;
  if (c < cols)
    for (int r = r0 + ty; r < r1; r += 8) {
      const float v = x[(size_t)r * ldx + c];
      a += (double)v;
      m = max(m, __float_as_uint(v) & 0x7fffffffu);
    }
  s[ty][tx] = a;
  if (amax != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  }
  __syncthreads();
  if (ty == 0 && c < cols) {
    double t = 0.0;
    for (int k = 0; k < 8; ++k) t += s[k][tx];
    part[(size_t)blockIdx.y * cols + c] = t;
  }
  if (amax != nullptr && threadIdx.x == 0)
    __hip_atomic_fetch_max(amax, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void col_sum_final_kernel(const double* __restrict__ part, int splits, int cols, float* __restrict__ out,
                                                            unsigned* __restrict__ amax, float* __restrict__ out2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (amax != nullptr && c == 0) {          // (every partial block has finished: this is a later launch of the same stream)
    const unsigned bits = __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float sc = 1.f;
    if (bits != 0 && bits < 0x7f800000u) sc = exp2f(10.f - floorf(log2f(__uint_as_float(bits))));
    out2[0] = sc;
    out2[1] = 1.f / sc;
    __hip_atomic_store(amax, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
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

// ---------------------------------------------------------------------------------------------- LayerNorm backward
// y = xhat * w + b over the last dimension (eps inside the sqrt), one wave per row (C <= 1280):
//   dx = rstd * (dz - mean(dz) - xhat * mean(dz * xhat)), dz = dy * w;  t = dy * xhat is written out so that dw = column sums of t and
//   db = column sums of dy come from mvd_col_sum (fixed order).
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                     int rows, int C, float eps, float* __restrict__ dx, float* __restrict__ t) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  const float* dr = dy + (size_t)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float a = xr[c] - mean;
    q += a * a;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  float m1 = 0.f, m2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mean) * rstd;
    const float dz = dr[c] * (w ? w[c] : 1.f);
    m1 += dz;
    m2 += dz * xh;
  }
  m1 = wave_sum(m1) / (float)C;
  m2 = wave_sum(m2) / (float)C;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mean) * rstd;
    const float dz = dr[c] * (w ? w[c] : 1.f);
    dx[(size_t)row * C + c] = rstd * (dz - m1 - xh * m2);
    if (t) t[(size_t)row * C + c] = dr[c] * xh;
  }
}

// ---------------------------------------------------------------------------------------------- GEGLU backward
// y = a * gelu(g) with [a | g] = h (rows, 2 * half) (attention.py:43-44, exact erf GELU):  dh = [dy * gelu(g) | dy * a * gelu'(g)],
// gelu'(g) = Phi(g) + g * phi(g).
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dy, int rows, int half,
                                                        float* __restrict__ dh) {
  const size_t total = (size_t)rows * half;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / half;
    const int c = (int)(e - r * half);
    const float a = h[r * 2 * half + c], g = h[r * 2 * half + half + c], d = dy[e];
    const float cdf = 0.5f * (1.0f + erf_nobranch(g * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
    dh[r * 2 * half + c] = d * (g * cdf);
    dh[r * 2 * half + half + c] = d * a * (cdf + g * pdf);
  }
}

// ---------------------------------------------------------------------------------------------- activations of the training step
// act_planes_kernel: y = act(x) (exact erf GELU as in the forward's epilogues, or SiLU) written as split planes and / or fp32 in ONE pass
// (the backward recomputes the DiT / pre-layer activations from the kept pre-activations: it was F.gelu + a split pass, 3 full-size
// tensors of traffic instead of 1 read + 1 write).  act_bwd_kernel: dx = dy * act'(x) in one pass (it was ~8 torch kernels per call:
// erf, exp, four multiplies, two adds over (T, 1024) tensors).  gelu'(x) = Phi(x) + x phi(x);  silu'(x) = s (1 + x (1 - s)).
template <int ACT>
__global__ __launch_bounds__(256) void act_planes_kernel(const float* __restrict__ x, u16* __restrict__ sp, float* __restrict__ y, size_t rows,
                                                         int cols, int ldx, int ldp, int ldy) {
  const int c4 = ldp >> 2;
  const size_t total = rows * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / c4;
    const int c = (int)(e - r * c4) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = (c + j) < cols ? x[r * ldx + c + j] : 0.f;
      v[j] = ACT == MVD_ACT_GELU ? gelu_erf(t) : t / (1.0f + __expf(-t));
    }
    if (sp != nullptr) store_sp4(sp, r, ldp, c, v[0], v[1], v[2], v[3]);
    if (y != nullptr) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < cols) y[r * ldy + c + j] = v[j];
    }
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, size_t n) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const float t = x[e], d = dy[e];
    float g;
    if (ACT == MVD_ACT_GELU) {
      const float cdf = 0.5f * (1.0f + erf_nobranch(t * 0.70710678118654752440f));
      g = cdf + t * (0.3989422804014327f * __expf(-0.5f * t * t));
    } else {
      const float sg = 1.0f / (1.0f + __expf(-t));
      g = sg * (1.0f + t * (1.0f - sg));
    }
    dx[e] = d * g;
  }
}

// ---------------------------------------------------------------------------------------------- self-attention backward (fp32 VALU)
// CrossAttention(context=None) core (attention.py:170-193): per (batch, head), O = softmax(Q K^T * scale) V over L tokens of width d.
// q, k, v, o-grad are token-major (B*L, heads*d) fp32 as the forward's projections produce them.  One LDS-tiled kernel, three modes:
//   MODE 2  stats : per query row  m = max_j s_ij, l = sum_j exp(s_ij - m), delta = sum_j P_ij dP_ij      (two sweeps over the keys)
//   MODE 0  dQ    : dQ_i = scale * sum_j P_ij (dP_ij - delta_i) K_j
//   MODE 1  dK,dV : dK_j = scale * sum_i P_ij (dP_ij - delta_i) Q_i,   dV_j = sum_i P_ij dO_i
// with P_ij = exp(s_ij - m_i) / l_i, dP_ij = dO_i . V_j.  A workgroup keeps TQ "stationary" rows (queries for modes 0 / 2, keys for
// mode 1: both of their operand rows) in LDS and streams tiles of TK rows of the other side; the 16 x 16 thread grid owns an
// (TQ/16) x (TK/16) block of the score tile, the dS / P tiles go through LDS for the second product.  Everything is accumulated in a
// fixed order: deterministic.  ~5 x 2 L^2 d FLOP per head on the VALU (the MFMA version would follow the forward's attn_kernel).
constexpr int AB_MAXD = 160;

template <int TQ, int TK, int DMAX, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_tile_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ dout,
                                                            float* __restrict__ stats, int L, int H, int d, float scale,
                                                            float* __restrict__ out1, float* __restrict__ out2) {
  constexpr int DP = DMAX + 1, RI = TQ / 16, RJ = TK / 16, EC = DMAX / 16;
  __shared__ float A1[TQ][DP], A2[TQ][DP], B1[TK][DP], B2[TK][DP];
  __shared__ float DSs[TQ][TK + 1];
  __shared__ float PSs[MODE == 1 ? TQ : 1][MODE == 1 ? TK + 1 : 1];
  __shared__ float stB[3][TK];             // mode 1: statistics of the streamed query rows
  __shared__ float red[2][TQ][17];         // mode 2: row reductions across the 16 column threads
  __shared__ float rowm[TQ];
  const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
  const int r0 = blockIdx.x * TQ, hd = blockIdx.y, b = blockIdx.z;
  const int C = H * d;
  const float* sA1 = MODE == 1 ? k : q;
  const float* sA2 = MODE == 1 ? v : dout;
  const float* sB1 = MODE == 1 ? q : k;
  const float* sB2 = MODE == 1 ? dout : v;
  const size_t base = (size_t)b * L * C + (size_t)hd * d;
  for (int idx = t; idx < TQ * d; idx += 256) {
    const int r = idx / d, e = idx - r * d;
    const bool ok = r0 + r < L;
    A1[r][e] = ok ? sA1[base + (size_t)(r0 + r) * C + e] : 0.f;
    A2[r][e] = ok ? sA2[base + (size_t)(r0 + r) * C + e] : 0.f;
  }
  float m_i[RI], il_i[RI], de_i[RI];
#pragma unroll
  for (int a = 0; a < RI; ++a) {
    m_i[a] = 0.f;
    il_i[a] = 0.f;
    de_i[a] = 0.f;
    if (MODE == 0) {
      const int row = r0 + ti * RI + a;
      if (row < L) {
        const float* st = stats + (((size_t)b * H + hd) * L + row) * 3;
        m_i[a] = st[0];
        il_i[a] = 1.0f / st[1];
        de_i[a] = st[2];
      }
    }
  }
  float acc1[RI][EC], acc2[RI][EC];
#pragma unroll
  for (int a = 0; a < RI; ++a)
#pragma unroll
    for (int c = 0; c < EC; ++c) acc1[a][c] = acc2[a][c] = 0.f;
  float mx[RI], ls[RI], dl[RI];
#pragma unroll
  for (int a = 0; a < RI; ++a) {
    mx[a] = -INFINITY;
    ls[a] = 0.f;
    dl[a] = 0.f;
  }
  const int sweeps = MODE == 2 ? 2 : 1;
  for (int sweep = 0; sweep < sweeps; ++sweep) {
    for (int c0 = 0; c0 < L; c0 += TK) {
      __syncthreads();
      for (int idx = t; idx < TK * d; idx += 256) {
        const int r = idx / d, e = idx - r * d;
        const bool ok = c0 + r < L;
        B1[r][e] = ok ? sB1[base + (size_t)(c0 + r) * C + e] : 0.f;
        B2[r][e] = ok ? sB2[base + (size_t)(c0 + r) * C + e] : 0.f;
      }
      if (MODE == 1 && t < TK) {
        const int row = c0 + t;
        const float* st = stats + (((size_t)b * H + hd) * L + (row < L ? row : 0)) * 3;
        stB[0][t] = st[0];
        stB[1][t] = 1.0f / st[1];
        stB[2][t] = st[2];
      }
      __syncthreads();
      // ---- score block: s = A1 . B1, dp = A2 . B2
      float s[RI][RJ], dp[RI][RJ];
#pragma unroll
      for (int a = 0; a < RI; ++a)
#pragma unroll
        for (int c = 0; c < RJ; ++c) s[a][c] = dp[a][c] = 0.f;
      for (int e = 0; e < d; ++e) {
        float a1[RI], a2[RI], b1[RJ], b2[RJ];
#pragma unroll
        for (int a = 0; a < RI; ++a) {
          a1[a] = A1[ti * RI + a][e];
          a2[a] = A2[ti * RI + a][e];
        }
#pragma unroll
        for (int c = 0; c < RJ; ++c) {
          b1[c] = B1[tj * RJ + c][e];
          b2[c] = B2[tj * RJ + c][e];
        }
#pragma unroll
        for (int a = 0; a < RI; ++a)
#pragma unroll
          for (int c = 0; c < RJ; ++c) {
            s[a][c] += a1[a] * b1[c];
            if (!(MODE == 2 && sweep == 0)) dp[a][c] += a2[a] * b2[c];
          }
      }
      if (MODE == 2) {
#pragma unroll
        for (int a = 0; a < RI; ++a)
#pragma unroll
          for (int c = 0; c < RJ; ++c) {
            const bool okc = c0 + tj * RJ + c < L;
            const float sv = s[a][c] * scale;
            if (sweep == 0) {
              if (okc) mx[a] = fmaxf(mx[a], sv);
            } else if (okc) {
              const float pu = __expf(sv - rowm[ti * RI + a]);
              ls[a] += pu;
              dl[a] += pu * dp[a][c];
            }
          }
        continue;
      }
      // ---- dS (and P) tiles -> LDS
#pragma unroll
      for (int a = 0; a < RI; ++a)
#pragma unroll
        for (int c = 0; c < RJ; ++c) {
          const int col = tj * RJ + c;
          const bool okc = c0 + col < L;
          const float m = MODE == 1 ? stB[0][col] : m_i[a];
          const float il = MODE == 1 ? stB[1][col] : il_i[a];
          const float de = MODE == 1 ? stB[2][col] : de_i[a];
          const float pr = okc ? __expf(s[a][c] * scale - m) * il : 0.f;
          DSs[ti * RI + a][col] = pr * (dp[a][c] - de) * scale;
          if (MODE == 1) PSs[ti * RI + a][col] = pr;
        }
      __syncthreads();
      // ---- second product: acc1[row][e] += sum_col dS[row][col] * B1[col][e]   (mode 1 also acc2 += P * B2)
#pragma unroll
      for (int a = 0; a < RI; ++a) {
        const int row = ti * RI + a;
        for (int col = 0; col < TK; ++col) {
          const float dsv = DSs[row][col];
          const float pv = MODE == 1 ? PSs[row][col] : 0.f;
#pragma unroll
          for (int c = 0; c < EC; ++c) {
            const int e = tj + 16 * c;
            acc1[a][c] += dsv * B1[col][e < DMAX ? e : 0];
            if (MODE == 1) acc2[a][c] += pv * B2[col][e < DMAX ? e : 0];
          }
        }
      }
    }
    if (MODE == 2) {   // reduce this sweep's row quantities across the 16 column threads (fixed order)
      __syncthreads();
#pragma unroll
      for (int a = 0; a < RI; ++a) {
        red[0][ti * RI + a][tj] = sweep == 0 ? mx[a] : ls[a];
        red[1][ti * RI + a][tj] = dl[a];
      }
      __syncthreads();
      if (t < TQ) {
        if (sweep == 0) {
          float mm = -INFINITY;
          for (int c = 0; c < 16; ++c) mm = fmaxf(mm, red[0][t][c]);
          rowm[t] = mm;
        } else {
          float l = 0.f, dd = 0.f;
          for (int c = 0; c < 16; ++c) {
            l += red[0][t][c];
            dd += red[1][t][c];
          }
          if (r0 + t < L) {
            float* st = stats + (((size_t)b * H + hd) * L + r0 + t) * 3;
            st[0] = rowm[t];
            st[1] = l;
            st[2] = dd / l;
          }
        }
      }
      __syncthreads();
    }
  }
  if (MODE == 2) return;
#pragma unroll
  for (int a = 0; a < RI; ++a) {
    const int row = r0 + ti * RI + a;
    if (row >= L) continue;
#pragma unroll
    for (int c = 0; c < EC; ++c) {
      const int e = tj + 16 * c;
      if (e < d) {
        out1[base + (size_t)row * C + e] = acc1[a][c];
        if (MODE == 1) out2[base + (size_t)row * C + e] = acc2[a][c];
      }
    }
  }
}

template <int TQ, int TK, int DMAX>
void launch_attn_bwd(const float* q, const float* k, const float* v, const float* dout, int B, int H, int L, int d, float scale, float* dq,
                     float* dk, float* dv, float* stats, hipStream_t s) {
  const dim3 grid((L + TQ - 1) / TQ, H, B);
  hipLaunchKernelGGL((attn_bwd_tile_kernel<TQ, TK, DMAX, 2>), grid, dim3(256), 0, s, q, k, v, dout, stats, L, H, d, scale, nullptr, nullptr);
  hipLaunchKernelGGL((attn_bwd_tile_kernel<TQ, TK, DMAX, 0>), grid, dim3(256), 0, s, q, k, v, dout, stats, L, H, d, scale, dq, nullptr);
  hipLaunchKernelGGL((attn_bwd_tile_kernel<TQ, TK, DMAX, 1>), grid, dim3(256), 0, s, q, k, v, dout, stats, L, H, d, scale, dk, dv);
}

// ---------------------------------------------------------------------------------------------- self-attention backward (fp32 MFMA)
// Round 6: the same mathematics as attn_bwd_tile_kernel above on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: fp32 operands, fp32
// accumulate -- no operand splitting, 157 TFLOP/s peak; the VALU kernel reached ~8).  Two kernels, 8 wavefronts per workgroup, a wavefront
// owns 16 "stationary" rows whose operands live in registers as MFMA B fragments; the other side streams through LDS in tiles of KT rows
// (fp32, odd pitch: conflict-free for both fragment patterns).  Every product is arranged so that the D layout of one MFMA (a lane holds rows
// 4g .. 4g + 3 of column c; g = lane / 16, c = lane % 16) IS the B fragment of the next (k-slot (g, i) <-> row 4g + i), the trick of the
// forward's attn_kernel: no score tile ever goes through LDS.
//   attn_bwd_q_mfma_kernel  (a wave = 16 queries): sweep 1 over the keys: S^T = K Q^T and dP^T = V dO^T per 16-key block, online
//       m_i, l_i = sum_j e^(s_ij - m_i), and sum_j e^(s_ij - m_i) dP_ij (rescaled like l) -> delta_i = sum_j P_ij dP_ij without O;
//       sweep 2: dS^T = P^T (dP^T - delta) scale, dQ^T += K^T dS^T; writes {m, l, delta} to `stats` for the second kernel;
//   attn_bwd_kv_mfma_kernel (a wave = 16 keys): streams the queries: S = Q K^T, dP = dO V^T, P from the stored statistics,
//       dV^T += dO^T P, dK^T += Q^T dS.
// Fixed summation order: deterministic.  d % 4 == 0, d <= 160; DB = 16-wide blocks of the head dim (zero padded in LDS).
template <int DB, int KT, int NT>
__device__ __forceinline__ void ab_stage_rows(float* tile, const float* __restrict__ src, int r0, int L, int C, int d, int tid) {
  constexpr int P = 16 * DB + 1;
  const int c4 = d >> 2;
  for (int idx = tid; idx < KT * c4; idx += NT) {
    const int r = idx / c4, c = (idx - r * c4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < L) v = *(const float4*)(src + (size_t)(r0 + r) * C + c);
    float* t = tile + r * P + c;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
}

template <int DB, int KT>
__global__ __launch_bounds__(512) void attn_bwd_q_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const float* __restrict__ dout, float* __restrict__ stats, int L, int H, int d,
                                                             float scale, float* __restrict__ dq) {
  constexpr int DPAD = 16 * DB, P = DPAD + 1, NT = 512, KS = DPAD / 4;
  __shared__ float sK[KT * P], sV[KT * P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int hd = blockIdx.y, b = blockIdx.z, C = H * d, ks = d >> 2;
  const size_t base = (size_t)b * L * C + (size_t)hd * d;
  const int qrow = blockIdx.x * 128 + wave * 16 + c;
  const bool qok = qrow < L;
  for (int i = tid; i < KT * P; i += NT) sK[i] = sV[i] = 0.f;      // (the padding columns [d, DPAD) stay zero)
  float qb[KS], dob[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int e = 4 * j + g;
    const bool ok = qok && e < d;
    qb[j] = ok ? q[base + (size_t)qrow * C + e] : 0.f;
    dob[j] = ok ? dout[base + (size_t)qrow * C + e] : 0.f;
  }
  f32x4 dqT[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) dqT[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f, dl_run = 0.f, inv_l = 0.f, delta = 0.f;
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (int kt0 = 0; kt0 < L; kt0 += KT) {
      __syncthreads();
      ab_stage_rows<DB, KT, NT>(sK, k + base, kt0, L, C, d, tid);
      ab_stage_rows<DB, KT, NT>(sV, v + base, kt0, L, C, d, tid);
      __syncthreads();
#pragma unroll 1
      for (int sub = 0; sub < KT / 16; ++sub) {
        if (kt0 + sub * 16 >= L) break;                     // (uniform: no key of this block exists)
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = s;
        const float* rk = sK + (sub * 16 + c) * P + g;
        const float* rv = sV + (sub * 16 + c) * P + g;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          if (j < ks) {
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(rk[4 * j], qb[j], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(rv[4 * j], dob[j], dp, 0, 0, 0);
          }
        }
        float sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i] = kt0 + sub * 16 + 4 * g + i < L ? s[i] * scale : -INFINITY;
        if (sweep == 0) {
          float mt = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
          mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
          mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
          const float m_new = fmaxf(m_run, mt);              // (finite: key kt0 + 16 sub exists)
          const float alpha = __expf(m_run - m_new);
          float ps = 0.f, pd = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pe = __expf(sv[i] - m_new);
            ps += pe;
            pd += pe * dp[i];
          }
          l_run = l_run * alpha + ps;
          dl_run = dl_run * alpha + pd;
          m_run = m_new;
        } else {
          const float* rkt = sK + (sub * 16 + 4 * g) * P + c;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pr = __expf(sv[i] - m_run) * inv_l;
            const float ds = pr * (dp[i] - delta) * scale;
#pragma unroll
            for (int blk = 0; blk < DB; ++blk) dqT[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rkt[i * P + blk * 16], ds, dqT[blk], 0, 0, 0);
          }
        }
      }
    }
    if (sweep == 0) {
      float lt = l_run + __shfl_xor(l_run, 16, 64);
      lt += __shfl_xor(lt, 32, 64);
      float dt = dl_run + __shfl_xor(dl_run, 16, 64);
      dt += __shfl_xor(dt, 32, 64);
      inv_l = 1.0f / lt;
      delta = dt * inv_l;
      if (g == 0 && qok) {
        float* st = stats + (((size_t)b * H + hd) * L + qrow) * 3;
        st[0] = m_run;
        st[1] = lt;
        st[2] = delta;
      }
    }
  }
  if (qok) {
#pragma unroll
    for (int blk = 0; blk < DB; ++blk) {
      const int dd = blk * 16 + 4 * g;
      if (dd < d) *(float4*)(dq + base + (size_t)qrow * C + dd) = make_float4(dqT[blk][0], dqT[blk][1], dqT[blk][2], dqT[blk][3]);
    }
  }
}

template <int DB, int KT>
__global__ __launch_bounds__(512) void attn_bwd_kv_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                              const float* __restrict__ dout, const float* __restrict__ stats, int L, int H, int d,
                                                              float scale, float* __restrict__ dk, float* __restrict__ dv) {
  constexpr int DPAD = 16 * DB, P = DPAD + 1, NT = 512, KS = DPAD / 4;
  __shared__ float sQ[KT * P], sO[KT * P], sS[KT * 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int hd = blockIdx.y, b = blockIdx.z, C = H * d, ks = d >> 2;
  const size_t base = (size_t)b * L * C + (size_t)hd * d;
  const int krow = blockIdx.x * 128 + wave * 16 + c;
  const bool kok = krow < L;
  for (int i = tid; i < KT * P; i += NT) sQ[i] = sO[i] = 0.f;
  float kb[KS], vb[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int e = 4 * j + g;
    const bool ok = kok && e < d;
    kb[j] = ok ? k[base + (size_t)krow * C + e] : 0.f;
    vb[j] = ok ? v[base + (size_t)krow * C + e] : 0.f;
  }
  f32x4 dkT[DB], dvT[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) dkT[i] = dvT[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* stb = stats + ((size_t)b * H + hd) * L * 3;
  for (int qt0 = 0; qt0 < L; qt0 += KT) {
    __syncthreads();
    ab_stage_rows<DB, KT, NT>(sQ, q + base, qt0, L, C, d, tid);
    ab_stage_rows<DB, KT, NT>(sO, dout + base, qt0, L, C, d, tid);
    for (int i = tid; i < KT * 3; i += NT) sS[i] = qt0 + i / 3 < L ? stb[(size_t)qt0 * 3 + i] : 0.f;
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < KT / 16; ++sub) {
      if (qt0 + sub * 16 >= L) break;
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = s;
      const float* rq = sQ + (sub * 16 + c) * P + g;
      const float* ro = sO + (sub * 16 + c) * P + g;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        if (j < ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x4f32(rq[4 * j], kb[j], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ro[4 * j], vb[j], dp, 0, 0, 0);
        }
      }
      const float* rqt = sQ + (sub * 16 + 4 * g) * P + c;
      const float* rot = sO + (sub * 16 + 4 * g) * P + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ql = sub * 16 + 4 * g + i;
        const bool ok = qt0 + ql < L;
        const float m = sS[ql * 3], l = sS[ql * 3 + 1], de = sS[ql * 3 + 2];
        const float pr = ok ? __expf(s[i] * scale - m) / l : 0.f;
        const float ds = pr * (dp[i] - de) * scale;
#pragma unroll
        for (int blk = 0; blk < DB; ++blk) {
          dvT[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rot[i * P + blk * 16], pr, dvT[blk], 0, 0, 0);
          dkT[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(rqt[i * P + blk * 16], ds, dkT[blk], 0, 0, 0);
        }
      }
    }
  }
  if (kok) {
#pragma unroll
    for (int blk = 0; blk < DB; ++blk) {
      const int dd = blk * 16 + 4 * g;
      if (dd < d) {
        *(float4*)(dk + base + (size_t)krow * C + dd) = make_float4(dkT[blk][0], dkT[blk][1], dkT[blk][2], dkT[blk][3]);
        *(float4*)(dv + base + (size_t)krow * C + dd) = make_float4(dvT[blk][0], dvT[blk][1], dvT[blk][2], dvT[blk][3]);
      }
    }
  }
}

template <int DB, int KT>
void launch_attn_bwd_mfma(const float* q, const float* k, const float* v, const float* dout, int B, int H, int L, int d, float scale, float* dq,
                          float* dk, float* dv, float* stats, hipStream_t s) {
  const dim3 grid((L + 127) / 128, H, B);
  hipLaunchKernelGGL((attn_bwd_q_mfma_kernel<DB, KT>), grid, dim3(512), 0, s, q, k, v, dout, stats, L, H, d, scale, dq);
  hipLaunchKernelGGL((attn_bwd_kv_mfma_kernel<DB, KT>), grid, dim3(512), 0, s, q, k, v, dout, stats, L, H, d, scale, dk, dv);
}

// ---------------------------------------------------------------------------------------------- per-pixel cross-attention backward
// DualAttnetionBlock.attn2 with D context tokens per pixel (mvdfusion/attention.py:52-62; forward: pixel_xattn_kernel): one wave per
// (pixel, head).  q (P, C), k / v (P*D, C), dout (P, C) -> dq (P, C), dk / dv (P*D, C).  D <= 8.
__global__ __launch_bounds__(256) void pixel_xattn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ dout, int P, int D,
                                                              int heads, int dhead, float* __restrict__ dq, float* __restrict__ dk,
                                                              float* __restrict__ dv) {
  const int lane = threadIdx.x & 63;
  const size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (size_t)P * heads) return;
  const size_t pix = item / heads;
  const int h = (int)(item - pix * heads);
  const int C = heads * dhead;
  const float scale = rsqrtf((float)dhead);
  const float* qr = q + pix * C + h * dhead;
  const float* dor = dout + pix * C + h * dhead;
  float sc[8], dp[8];
  float mx = -INFINITY;
  for (int j = 0; j < D; ++j) {
    const float* kr = k + (pix * D + j) * C + h * dhead;
    const float* vr = v + (pix * D + j) * C + h * dhead;
    float a = 0.f, c = 0.f;
    for (int e = lane; e < dhead; e += 64) {
      a += qr[e] * kr[e];
      c += dor[e] * vr[e];
    }
    sc[j] = wave_sum(a) * scale;
    dp[j] = wave_sum(c);
    mx = fmaxf(mx, sc[j]);
  }
  float den = 0.f;
  for (int j = 0; j < D; ++j) {
    sc[j] = expf(sc[j] - mx);
    den += sc[j];
  }
  float delta = 0.f;
  for (int j = 0; j < D; ++j) {
    sc[j] /= den;
    delta += sc[j] * dp[j];
  }
  for (int e = lane; e < dhead; e += 64) {
    float gq = 0.f;
    for (int j = 0; j < D; ++j) {
      const size_t rj = (pix * D + j) * C + h * dhead + e;
      const float ds = sc[j] * (dp[j] - delta) * scale;
      gq += ds * k[rj];
      dk[rj] = ds * qr[e];
      dv[rj] = sc[j] * dor[e];
    }
    dq[pix * C + h * dhead + e] = gq;
  }
}

// ------------------------------------------------------------------------------------------------ gradient scale / optimizer
// mvd_pow2_scale: out = {s, 1/s}, s the power of two that brings max|x| to [1024, 2048) (1 when the maximum is 0 or not finite).  One
// launch: block maxima meet in an atomicMax on the float's bit pattern (non-negative floats order like their bits; NaN and inf sort above
// every finite value and are caught below), the last block to arrive forms the scale and re-zeroes the two scratch words for the next call.
__global__ __launch_bounds__(256) void pow2_scale_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out, unsigned* scratch) {
  unsigned m = 0;
  const size_t n4 = n >> 2;
  const float4* x4 = (const float4*)x;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = x4[i];
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, __float_as_uint(v.z) & 0x7fffffffu));
    m = max(m, __float_as_uint(v.w) & 0x7fffffffu);
  }
  if (blockIdx.x == 0)
    for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  __shared__ unsigned s_m[4];
  __shared__ bool s_last;
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    __hip_atomic_fetch_max(&scratch[0], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = __hip_atomic_fetch_add(&scratch[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    if (s_last) {
      const unsigned bits = __hip_atomic_load(&scratch[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float s = 1.f;
      if (bits != 0 && bits < 0x7f800000u) s = exp2f(10.f - floorf(log2f(__uint_as_float(bits))));
      out[0] = s;
      out[1] = 1.f / s;
      __hip_atomic_store(&scratch[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&scratch[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// mvd_adamw_multi: torch.optim.AdamW's update (decoupled weight decay, bias correction, no amsgrad) for a LIST of tensors in one launch.
// Table: per tensor {p, g, m, v, numel, first_chunk}; block b owns chunk b of 4096 elements of the concatenated index space and finds
// its tensor by bisection over first_chunk (uniform: scalar loads).
struct AdamwTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  unsigned long long numel;
  unsigned first_chunk;      // index of this tensor's first 4096-element chunk
  unsigned pad;
};
constexpr int ADAMW_CHUNK = 4096;
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwTensor* __restrict__ tab, int n_tensors, float lr, float beta1, float beta2,
                                                          float eps, float weight_decay, float bias_c1, float bias_c2_sqrt, float grad_scale) {
  int lo = 0, hi = n_tensors - 1;
  while (lo < hi) {                                   // last tensor whose first chunk is <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_chunk <= blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const AdamwTensor t = tab[lo];
  const size_t base = (size_t)(blockIdx.x - t.first_chunk) * ADAMW_CHUNK;
  const size_t left = t.numel - base;
  const int n = left < (size_t)ADAMW_CHUNK ? (int)left : ADAMW_CHUNK;
  const float step_size = lr / bias_c1, decay = 1.f - lr * weight_decay, omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  auto upd = [&](float& p, float g, float& m, float& v) {
    g *= grad_scale;
    p *= decay;                                       // param.mul_(1 - lr * weight_decay)
    m = m + (g - m) * omb1;                           // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + omb2 * g * g;                     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / bias_c2_sqrt + eps;
    p -= step_size * (m / denom);                     // param.addcdiv_(exp_avg, denom, value = -step_size)
  };
  float* pp = t.p + base;
  const float* gp = t.g + base;
  float* mp = t.m + base;
  float* vp = t.v + base;
  const bool vec = ((((uintptr_t)pp | (uintptr_t)gp | (uintptr_t)mp | (uintptr_t)vp) & 15) == 0);
  const int n4 = vec ? n >> 2 : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 p = ((float4*)pp)[i], m = ((float4*)mp)[i], v = ((float4*)vp)[i];
    const float4 g = ((const float4*)gp)[i];
    upd(p.x, g.x, m.x, v.x);
    upd(p.y, g.y, m.y, v.y);
    upd(p.z, g.z, m.z, v.z);
    upd(p.w, g.w, m.w, v.w);
    ((float4*)pp)[i] = p;
    ((float4*)mp)[i] = m;
    ((float4*)vp)[i] = v;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
    float p = pp[i], m = mp[i], v = vp[i];
    upd(p, gp[i], m, v);
    pp[i] = p;
    mp[i] = m;
    vp[i] = v;
  }
}

}  // namespace

static int transpose_planes_impl(const void* x, int src_planes, int rows, int cols, int ldx, void* out_sp, int ldo, const float* scale_dev,
                                 mvd_stream_t stream) {
  MVD_CHECK_ARG(x && out_sp && rows > 0 && cols > 0 && ldx >= cols, "mvd_transpose_planes: bad arguments");
  MVD_CHECK_ARG(ldo % 32 == 0 && ldo >= (rows + 31) / 32 * 32 && ((uintptr_t)out_sp & 127) == 0,
                "mvd_transpose_planes: ldo=%d must be a multiple of 32 covering %d rows, 128-byte aligned output", ldo, rows);
  if (src_planes) MVD_CHECK_ARG(ldx % 32 == 0, "mvd_transpose_planes: plane source needs ldx %% 32 == 0");
  hipLaunchKernelGGL(transpose_planes_kernel, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, src_planes,
                     rows, cols, ldx, (u16*)out_sp, ldo, scale_dev);
  MVD_CHECK_LAUNCH("mvd_transpose_planes");
  return 0;
}

extern "C" int mvd_transpose_planes(const void* x, int src_planes, int rows, int cols, int ldx, void* out_sp, int ldo, mvd_stream_t stream) {
  return transpose_planes_impl(x, src_planes, rows, cols, ldx, out_sp, ldo, nullptr, stream);
}

extern "C" int mvd_transpose_planes_scaled(const float* x, int rows, int cols, int ldx, void* out_sp, int ldo, const float* scale_dev,
                                           mvd_stream_t stream) {
  return transpose_planes_impl(x, 0, rows, cols, ldx, out_sp, ldo, scale_dev, stream);
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

static int col_sum_launch(const float* x, int rows, int cols, int ldx, float* out, double* ws, size_t ws_doubles, unsigned* amax, float* out2,
                          hipStream_t stream) {
  int splits = rows / 256;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  MVD_CHECK_ARG(ws_doubles >= (size_t)splits * cols, "mvd_col_sum: workspace too small (%zu < %zu doubles)", ws_doubles,
                (size_t)splits * cols);
  hipLaunchKernelGGL(col_sum_partial_kernel, dim3((cols + 31) / 32, splits), dim3(256), 0, stream, x, rows, cols, ldx, ws, amax);
  hipLaunchKernelGGL(col_sum_final_kernel, dim3((cols + 255) / 256), dim3(256), 0, stream, ws, splits, cols, out, amax, out2);
  MVD_CHECK_LAUNCH("mvd_col_sum");
  return 0;
}

extern "C" int mvd_col_sum(const float* x, int rows, int cols, int ldx, float* out, double* ws, size_t ws_doubles, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && out && ws && rows > 0 && cols > 0 && ldx >= cols, "mvd_col_sum: bad arguments");
  return col_sum_launch(x, rows, cols, ldx, out, ws, ws_doubles, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int mvd_col_sum_pow2(const float* x, int rows, int cols, int ldx, float* out, double* ws, size_t ws_doubles, float* out2,
                                unsigned* scratch1, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && out && ws && out2 && scratch1 && rows > 0 && cols > 0 && ldx >= cols, "mvd_col_sum_pow2: bad arguments");
  return col_sum_launch(x, rows, cols, ldx, out, ws, ws_doubles, scratch1, out2, (hipStream_t)stream);
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

extern "C" int mvd_layernorm_backward(const float* x, const float* dy, const float* w, int rows, int C, float eps, float* dx, float* dyxhat,
                                      mvd_stream_t stream) {
  MVD_CHECK_ARG(x && dy && dx && rows > 0 && C > 0, "mvd_layernorm_backward: bad arguments");
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, w, rows, C, eps, dx, dyxhat);
  MVD_CHECK_LAUNCH("mvd_layernorm_backward");
  return 0;
}

extern "C" int mvd_geglu_backward(const float* h, const float* dy, int rows, int half, float* dh, mvd_stream_t stream) {
  MVD_CHECK_ARG(h && dy && dh && rows > 0 && half > 0, "mvd_geglu_backward: bad arguments");
  const size_t total = (size_t)rows * half;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, h, dy, rows, half, dh);
  MVD_CHECK_LAUNCH("mvd_geglu_backward");
  return 0;
}

extern "C" int mvd_act_planes(const float* x, void* sp, float* y, size_t rows, int cols, int ldx, int ldp, int ldy, int act, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && (sp || y) && rows > 0 && cols > 0 && ldx >= cols && ldp >= cols && ldp % 32 == 0 && (!y || ldy >= cols),
                "mvd_act_planes: bad arguments");
  MVD_CHECK_ARG(act == MVD_ACT_GELU || act == MVD_ACT_SILU, "mvd_act_planes: act %d (GELU or SiLU)", act);
  const size_t total = rows * (size_t)(ldp >> 2);
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (act == MVD_ACT_GELU)
    hipLaunchKernelGGL(act_planes_kernel<MVD_ACT_GELU>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (u16*)sp, y, rows, cols, ldx, ldp, ldy);
  else
    hipLaunchKernelGGL(act_planes_kernel<MVD_ACT_SILU>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (u16*)sp, y, rows, cols, ldx, ldp, ldy);
  MVD_CHECK_LAUNCH("mvd_act_planes");
  return 0;
}

extern "C" int mvd_act_backward(const float* dy, const float* x, float* dx, size_t n, int act, mvd_stream_t stream) {
  MVD_CHECK_ARG(dy && x && dx && n > 0, "mvd_act_backward: bad arguments");
  MVD_CHECK_ARG(act == MVD_ACT_GELU || act == MVD_ACT_SILU, "mvd_act_backward: act %d (GELU or SiLU)", act);
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (act == MVD_ACT_GELU) hipLaunchKernelGGL(act_bwd_kernel<MVD_ACT_GELU>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n);
  else hipLaunchKernelGGL(act_bwd_kernel<MVD_ACT_SILU>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n);
  MVD_CHECK_LAUNCH("mvd_act_backward");
  return 0;
}

extern "C" int mvd_attention_backward(const float* q, const float* k, const float* v, const float* dout, int B, int heads, int L, int dhead,
                                      float* dq, float* dk, float* dv, float* stats, size_t stats_floats, mvd_stream_t stream) {
  MVD_CHECK_ARG(q && k && v && dout && dq && dk && dv && stats, "mvd_attention_backward: null pointer");
  MVD_CHECK_ARG(B > 0 && heads > 0 && L > 0 && dhead > 0 && dhead % 4 == 0 && dhead <= AB_MAXD,
                "mvd_attention_backward: dhead=%d must be a multiple of 4 and <= %d", dhead, AB_MAXD);
  MVD_CHECK_ARG(stats_floats >= (size_t)B * heads * L * 3, "mvd_attention_backward: stats workspace too small");
  MVD_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout) & 15) == 0, "mvd_attention_backward: 16-byte alignment");
  MVD_CHECK_ARG(B <= 65535 && heads <= 65535, "mvd_attention_backward: batch / heads exceed the grid limits");
  const float scale = 1.0f / sqrtf((float)dhead);
  hipStream_t s = (hipStream_t)stream;
  // MVD_ATTN_BWD_VALU=1: the round-5 fp32 VALU kernels for every shape (A/B runs, cross-check in tests/test_gpu_backward.py)
  static const bool valu_only = getenv("MVD_ATTN_BWD_VALU") != nullptr && getenv("MVD_ATTN_BWD_VALU")[0] == '1';
  if (L <= 16 && dhead <= 48)            // the sequences over the V reference views of GridAttn (huge batch of tiny problems): VALU
    launch_attn_bwd<16, 16, 48>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  else if (!valu_only && dhead <= 16)
    launch_attn_bwd_mfma<1, 64>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  else if (!valu_only && dhead <= 48)
    launch_attn_bwd_mfma<3, 64>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  else if (!valu_only && dhead <= 80)
    launch_attn_bwd_mfma<5, 64>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  else if (!valu_only)
    launch_attn_bwd_mfma<10, 32>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  else if (dhead <= 48)
    launch_attn_bwd<64, 32, 48>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  else if (dhead <= 96)
    launch_attn_bwd<32, 32, 96>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  else
    launch_attn_bwd<16, 16, 160>(q, k, v, dout, B, heads, L, dhead, scale, dq, dk, dv, stats, s);
  MVD_CHECK_LAUNCH("mvd_attention_backward");
  return 0;
}

extern "C" int mvd_pixel_cross_attn_backward(const float* q, const float* k, const float* v, const float* dout, int P, int D, int heads,
                                             int dhead, float* dq, float* dk, float* dv, mvd_stream_t stream) {
  MVD_CHECK_ARG(q && k && v && dout && dq && dk && dv && P > 0 && D > 0 && D <= 8 && heads > 0 && dhead > 0,
                "mvd_pixel_cross_attn_backward: bad arguments (D <= 8)");
  const size_t items = (size_t)P * heads;
  hipLaunchKernelGGL(pixel_xattn_bwd_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, k, v, dout, P, D,
                     heads, dhead, dq, dk, dv);
  MVD_CHECK_LAUNCH("mvd_pixel_cross_attn_backward");
  return 0;
}


extern "C" int mvd_pow2_scale(const float* x, size_t n, float* out2, unsigned* scratch2, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && out2 && scratch2 && n > 0 && ((uintptr_t)x & 15) == 0, "mvd_pow2_scale: null / misaligned argument");
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(pow2_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out2, scratch2);
  MVD_CHECK_LAUNCH("mvd_pow2_scale");
  return 0;
}

extern "C" int mvd_adamw_multi(const void* tensors, int n_tensors, int n_chunks, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float bias_c1, float bias_c2_sqrt, float grad_scale, mvd_stream_t stream) {
  MVD_CHECK_ARG(tensors != nullptr && n_tensors >= 0 && n_chunks >= 0, "mvd_adamw_multi: null table");
  if (n_tensors == 0 || n_chunks == 0) return 0;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamwTensor*)tensors, n_tensors, lr, beta1,
                     beta2, eps, weight_decay, bias_c1, bias_c2_sqrt, grad_scale);
  MVD_CHECK_LAUNCH("mvd_adamw_multi");
  return 0;
}
