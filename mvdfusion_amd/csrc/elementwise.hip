// Small HBM/latency-bound kernels: fp32 GEMV (M <= 16), layout conversion, concat, area pooling, per-step scalars,
// CFG combine + DDIM update.  All exact fp32 VALU math.
#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == MVD_ACT_GELU) return gelu_erf(v);
  if (act == MVD_ACT_SILU) return silu_f(v);
  return v;
}

// one wave per output feature n; lanes stride over K (float4); up to 16 rows of x share one sweep of W.
template <int MR>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                   const float* __restrict__ x, float* __restrict__ y, int M, int N, int K, int ldx,
                                                   int ldy, int act_in, int act_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  const float* wr = W + (size_t)n * K;
  if ((K & 3) == 0) {
    for (int k = lane * 4; k < K; k += 256) {
      const float4 w4 = *(const float4*)(wr + k);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        if (m < M) {
          float4 x4 = *(const float4*)(x + (size_t)m * ldx + k);
          if (act_in) {
            x4.x = act_apply(x4.x, act_in);
            x4.y = act_apply(x4.y, act_in);
            x4.z = act_apply(x4.z, act_in);
            x4.w = act_apply(x4.w, act_in);
          }
          acc[m] += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
        }
      }
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      const float w1 = wr[k];
#pragma unroll
      for (int m = 0; m < MR; ++m)
        if (m < M) acc[m] += w1 * act_apply(x[(size_t)m * ldx + k], act_in);
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m < M) {
      float v = wave_sum(acc[m]);
      if (lane == 0) {
        if (bias) v += bias[n];
        y[(size_t)m * ldy + n] = act_apply(v, act_out);
      }
    }
  }
}

__global__ __launch_bounds__(256) void unet_input_kernel(const float* __restrict__ x, const float* __restrict__ il,
                                                         u16* __restrict__ out_sp, int V, int S, int cpad, int nb) {
  const int SS = S * S;
  const size_t total = (size_t)nb * SS * cpad;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % cpad);
    const int pix = (int)((e / cpad) % SS);
    const int b = (int)(e / ((size_t)cpad * SS));
    const int v = b < V ? b : b - V;
    float o = 0.f;
    if (c < 5) {
      o = x[((size_t)v * 5 + c) * SS + pix];
    } else if (c < 10 && b < V) {
      const float t = il[(size_t)(c - 5) * SS + pix];
      o = (c < 9) ? t / 0.18215f : t;
    }
    store_sp1(out_sp, e / cpad, cpad, c, o);
  }
}

__global__ __launch_bounds__(256) void concat_kernel(const float4* __restrict__ a, int ca4, const float4* __restrict__ b, int cb4,
                                                     float4* __restrict__ out, u16* __restrict__ out_sp, size_t rows) {
  const int c4 = ca4 + cb4;
  const size_t total = rows * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / c4;
    const int c = (int)(e - r * c4);
    const float4 v = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
    out[e] = v;
    if (out_sp) store_sp4(out_sp, r, c4 * 4, c * 4, v.x, v.y, v.z, v.w);
  }
}

// concat + GroupNorm statistics of the result: one workgroup per 16-row slab, threads over the float4 columns, rows in order
__global__ __launch_bounds__(256) void concat_stats_kernel(const float4* __restrict__ a, int ca4, const float4* __restrict__ b, int cb4,
                                                           float4* __restrict__ out, u16* __restrict__ out_sp, long long* __restrict__ stats,
                                                           int hw, int groups) {
  // workgroup = (16-row slab, 256-column span); thread = (4 rows, one float4 column) -- see splitk_reduce_stats_kernel (gemm.hip)
  __shared__ float cs[2][4][256];
  const int c4n = ca4 + cb4, C = c4n * 4;
  const size_t m0 = (size_t)blockIdx.x * 16;
  const int n0 = blockIdx.y * 256;
  const int c4 = blockIdx.y * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < c4n) {
    float4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = m0 + rg * 4 + r;
      v[r] = c4 < ca4 ? a[row * ca4 + c4] : b[row * cb4 + (c4 - ca4)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = m0 + rg * 4 + r;
      out[row * c4n + c4] = v[r];
      if (out_sp) store_sp4(out_sp, row, C, c4 * 4, v[r].x, v[r].y, v[r].z, v[r].w);
      s.x += v[r].x; s.y += v[r].y; s.z += v[r].z; s.w += v[r].w;
      q.x += v[r].x * v[r].x; q.y += v[r].y * v[r].y; q.z += v[r].z * v[r].z; q.w += v[r].w * v[r].w;
    }
  }
  *(float4*)&cs[0][rg][(threadIdx.x & 63) * 4] = s;
  *(float4*)&cs[1][rg][(threadIdx.x & 63) * 4] = q;
  __syncthreads();
  const int c = threadIdx.x, nn = n0 + c;
  const float cs_ = cs[0][0][c] + cs[0][1][c] + cs[0][2][c] + cs[0][3][c];
  const float cq_ = cs[1][0][c] + cs[1][1][c] + cs[1][2][c] + cs[1][3][c];
  __syncthreads();
  cs[0][0][c] = cs_;
  cs[1][0][c] = cq_;
  __syncthreads();
  if (nn >= C) return;
  const int cg = C / groups;
  const int g = nn / cg, pos = nn - g * cg;
  if (c != 0 && pos != 0) return;
  int len = cg - pos;
  if (len > 256 - c) len = 256 - c;
  float ss = 0.f, qq = 0.f;
  for (int j = 0; j < len; ++j) {
    ss += cs[0][0][c + j];
    qq += cs[1][0][c + j];
  }
  gn_stats_add(stats, (int)(m0 / hw), g, groups, ss, qq);
}

// concat + GroupNorm (+ SiLU) of the result in ONE launch (unet.py:550 torch.cat([h, hs.pop()]) -> ResBlock in_layers, openaimodel.py:201-204):
// a workgroup per (image, group) gathers its hw x (C / groups) values from the two sources into LDS, forms mean / rstd (fp32 partials per
// thread in a fixed order, combined in double) and writes the normalised planes, the RAW planes (the ResBlock's 1x1 skip convolution
// reads those) and, on request, the fp32 concatenation -- what concat_stats_kernel + gn_apply_stats(_cols)_kernel did in two launches.
// Same structure as splitk_gn_kernel (gemm.hip): two channels per thread, U elements' loads in flight at a time.
#define MVD_CGN_THREADS 1024
__global__ __launch_bounds__(MVD_CGN_THREADS) void concat_gn_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb,
                                                                    float* __restrict__ out, u16* __restrict__ raw_sp, u16* __restrict__ y_sp,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    long long* __restrict__ stats, int hw, int groups, float eps, int flags) {
  constexpr int NT = MVD_CGN_THREADS, NWV = NT / 64, U = 4;
  extern __shared__ float s_val[];
  __shared__ double s_red[2][NWV];
  __shared__ float s_coef[2][128];
  const int C = ca + cb, cg = C / groups, cg2 = cg >> 1;
  int g, img;
  {
    const int bid = blockIdx.x;
    if ((groups & 7) == 0) {
      const int gpx = groups >> 3, x = bid & 7, r = bid >> 3;
      g = x * gpx + r % gpx;
      img = r / gpx;
    } else {
      g = bid % groups;
      img = bid / groups;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = g * cg;
  const size_t m0 = (size_t)img * hw;
  const int total = hw * cg2;
  const int dq = NT / cg2, dj = NT - dq * cg2;
  const int rs = tid / cg2, js = tid - rs * cg2;
  const int r0 = tid < total ? rs : 0, j0 = tid < total ? js : 0;
  float s = 0.f, q = 0.f;
  {
    int r = rs, j = js;
    for (int e0 = tid; e0 < total; e0 += NT * U) {
      float2 v[U];
      int rr[U], jj[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = e0 + u * NT < total;
        rr[u] = ok[u] ? r : r0;
        jj[u] = ok[u] ? j : j0;
        r += dq;
        j += dj;
        if (j >= cg2) {
          j -= cg2;
          ++r;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t m = m0 + rr[u];
        const int n = c0 + 2 * jj[u];
        v[u] = *(const float2*)(n < ca ? a + m * ca + n : b + m * cb + (n - ca));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          const size_t m = m0 + rr[u];
          const int n = c0 + 2 * jj[u];
          if (out) *(float2*)(out + m * C + n) = v[u];
          if (raw_sp) {
            uint32_t hh, ll;
            split_op16x2(v[u].x, v[u].y, hh, ll);
            u16* pp = raw_sp + sp_index(m, C, n);
            *(uint32_t*)pp = hh;
            *(uint32_t*)(pp + 32) = ll;
          }
          *(float2*)(s_val + 2 * (e0 + u * NT)) = v[u];
          s += v[u].x + v[u].y;
          q += v[u].x * v[u].x + v[u].y * v[u].y;
        }
      }
    }
  }
  {
    const double sd = wave_sum_d((double)s), qd = wave_sum_d((double)q);
    if (lane == 0) {
      s_red[0][wave] = sd;
      s_red[1][wave] = qd;
    }
  }
  __syncthreads();
  double S1 = 0.0, S2 = 0.0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) {
    S1 += s_red[0][w];
    S2 += s_red[1][w];
  }
  const double cnt = (double)hw * cg;
  const double mean = S1 / cnt;
  double var = S2 / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (tid < cg) {
    const float aa = rstd * gamma[c0 + tid];
    s_coef[0][tid] = aa;
    s_coef[1][tid] = beta[c0 + tid] - (float)mean * aa;
  }
  if (tid == 0 && stats) gn_stats_add(stats, img, g, groups, (float)S1, (float)S2);
  __syncthreads();
  {
    int r = rs, j = js;
    for (int e = tid; e < total; e += NT) {
      float2 v = *(const float2*)(s_val + 2 * e);
      v.x = v.x * s_coef[0][2 * j] + s_coef[1][2 * j];
      v.y = v.y * s_coef[0][2 * j + 1] + s_coef[1][2 * j + 1];
      if (flags & 2) {
        v.x = (float)(_Float16)v.x;
        v.y = (float)(_Float16)v.y;
      }
      if (flags & 1) {
        v.x = silu_f(v.x);
        v.y = silu_f(v.y);
      }
      uint32_t hh, ll;
      split_op16x2(v.x, v.y, hh, ll);
      u16* pp = y_sp + sp_index(m0 + r, C, c0 + 2 * j);
      *(uint32_t*)pp = hh;
      *(uint32_t*)(pp + 32) = ll;
      r += dq;
      j += dj;
      if (j >= cg2) {
        j -= cg2;
        ++r;
      }
    }
  }
}

// vol (B,S,S,D,C) -> out (B,S/f,S/f,D,C), mean over f x f windows (F.interpolate(mode='area') with integer ratio)
__global__ __launch_bounds__(256) void area_pool_kernel(const float4* __restrict__ vol, u16* __restrict__ out_sp, int B, int S,
                                                        int D, int C4, int f, int ldp) {
  const int So = S / f;
  const size_t total = (size_t)B * So * So * D * C4;
  const float inv = 1.0f / (float)(f * f);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C4);
    size_t r = e / C4;
    const int d = (int)(r % D);
    r /= D;
    const int ox = (int)(r % So);
    r /= So;
    const int oy = (int)(r % So);
    const int b = (int)(r / So);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < f; ++dy)
#pragma unroll 8
      for (int dx = 0; dx < f; ++dx) {      // (unrolled: the window's loads of a row go out together; the additions keep their order)
        const float4 v = vol[((((size_t)b * S + oy * f + dy) * S + ox * f + dx) * D + d) * C4 + c];
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
      }
    store_sp4(out_sp, e / C4, ldp, c * 4, acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

__global__ __launch_bounds__(256) void fill_zero_kernel(float* p, size_t n) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) p[e] = 0.f;
}

__global__ void timestep_embedding_kernel(const float* __restrict__ steps, const int* __restrict__ iter,
                                          const float* __restrict__ freqs, float* __restrict__ out, int dim) {
  const float t = steps[(size_t)iter[0] * MVD_STEP_STRIDE + 0];
  const int half = dim / 2;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float a = t * freqs[i];
    out[i] = cosf(a);
    out[half + i] = sinf(a);
  }
}

__global__ void advance_iter_kernel(int* iter) {
  if (threadIdx.x == 0) iter[0] += 1;
}

__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* __restrict__ eps_nhwc, int ldc, float* __restrict__ x,
                                                       float* __restrict__ x0, float* __restrict__ eps_out,
                                                       const float* __restrict__ noise, const float* __restrict__ steps,
                                                       const int* __restrict__ iter, int V, int S, int cfg, float cfg_scale,
                                                       int do_update, size_t noise_stride) {
  const int it = iter[0];
  const float* st = steps + (size_t)it * MVD_STEP_STRIDE;
  const float a_t = st[3], a_prev = st[4], sigma = st[5], s1m = st[6], has_noise = st[7];
  const int SS = S * S;
  const size_t total = (size_t)V * 5 * SS;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int pix = (int)(e % SS);
    const int c = (int)((e / SS) % 5);
    const int b = (int)(e / ((size_t)5 * SS));
    float ep = eps_nhwc[((size_t)b * SS + pix) * ldc + c];
    if (cfg) {
      const float eu = eps_nhwc[((size_t)(V + b) * SS + pix) * ldc + c];
      ep = eu + cfg_scale * (ep - eu);
    }
    if (eps_out) eps_out[e] = ep;
    if (do_update) {
      const float xv = x[e];
      const float px0 = (xv - s1m * ep) / sqrtf(a_t);
      const float dir = sqrtf(fmaxf(1.0f - a_prev - sigma * sigma, 1e-7f)) * ep;
      float xp = sqrtf(a_prev) * px0 + dir;
      if (has_noise != 0.f) xp = xp + sigma * noise[(size_t)it * noise_stride + e];
      x0[e] = px0;
      x[e] = xp;
    }
  }
}

inline int grid_for(size_t total, int cap = 2048) {
  size_t b = (total + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int mvd_gemv(const float* W, const float* bias, const float* x, float* y, int M, int N, int K, int ldx, int ldy,
                        int act_in, int act_out, mvd_stream_t stream) {
  MVD_CHECK_ARG(W && x && y && M > 0 && M <= 16 && N > 0 && K > 0, "mvd_gemv: bad arguments (M <= 16)");
  if ((K & 3) == 0) MVD_CHECK_ARG((ldx & 3) == 0, "mvd_gemv: ldx must be a multiple of 4 when K is");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(N, 4)), block(256);
  if (M <= 1)
    hipLaunchKernelGGL(gemv_kernel<1>, grid, block, 0, s, W, bias, x, y, M, N, K, ldx, ldy, act_in, act_out);
  else if (M <= 4)
    hipLaunchKernelGGL(gemv_kernel<4>, grid, block, 0, s, W, bias, x, y, M, N, K, ldx, ldy, act_in, act_out);
  else
    hipLaunchKernelGGL(gemv_kernel<16>, grid, block, 0, s, W, bias, x, y, M, N, K, ldx, ldy, act_in, act_out);
  MVD_CHECK_LAUNCH("mvd_gemv");
  return 0;
}

extern "C" int mvd_unet_input(const float* x, const float* input_latents, void* out_sp, int V, int S, int cpad, int cfg,
                              mvd_stream_t stream) {
  MVD_CHECK_ARG(x && input_latents && out_sp && cpad % 32 == 0 && V > 0 && S > 0 && cpad >= 10, "mvd_unet_input: bad arguments");
  const int nb = cfg ? 2 * V : V;
  const size_t total = (size_t)nb * S * S * cpad;
  hipLaunchKernelGGL(unet_input_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, input_latents, (u16*)out_sp, V,
                     S, cpad, nb);
  MVD_CHECK_LAUNCH("mvd_unet_input");
  return 0;
}

extern "C" int mvd_concat_channels(const float* a, int Ca, const float* b, int Cb, float* out, void* out_sp, int rows,
                                   long long* gn_stats, int gn_hw, int gn_groups, mvd_stream_t stream) {
  if (out_sp) MVD_CHECK_ARG((Ca + Cb) % 32 == 0, "mvd_concat_channels: split-planes output needs (Ca+Cb) %% 32 == 0");
  MVD_CHECK_ARG(a && b && out && rows > 0 && Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0,
                "mvd_concat_channels: bad arguments (channel counts must be multiples of 4)");
  if (gn_stats) {
    MVD_CHECK_ARG(rows % 16 == 0 && gn_hw > 0 && gn_hw % 16 == 0 && gn_groups > 0 && (Ca + Cb) % gn_groups == 0,
                  "mvd_concat_channels: gn_stats needs rows %% 16 == 0, gn_hw %% 16 == 0, C %% groups == 0");
    hipLaunchKernelGGL(concat_stats_kernel, dim3(rows / 16, (Ca + Cb + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float4*)a, Ca / 4,
                       (const float4*)b, Cb / 4, (float4*)out, (u16*)out_sp, gn_stats, gn_hw, gn_groups);
    MVD_CHECK_LAUNCH("mvd_concat_channels/stats");
    return 0;
  }
  const size_t total = (size_t)rows * (Ca + Cb) / 4;
  hipLaunchKernelGGL(concat_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, Ca / 4,
                     (const float4*)b, Cb / 4, (float4*)out, (u16*)out_sp, (size_t)rows);
  MVD_CHECK_LAUNCH("mvd_concat_channels");
  return 0;
}

extern "C" int mvd_concat_groupnorm_fits(int Ca, int Cb, int hw, int groups) {
  const int C = Ca + Cb;
  if (groups <= 0 || C % groups || (Ca & 1) || (Cb & 1)) return 0;
  const int cg = C / groups;
  return (cg & 1) == 0 && cg <= 128 && (size_t)hw * cg * 4 <= 128 * 1024 ? 1 : 0;
}

extern "C" int mvd_concat_groupnorm(const float* a, int Ca, const float* b, int Cb, float* out, void* raw_sp, void* y_sp, const float* gamma,
                                    const float* beta, long long* gn_stats, int B, int hw, int groups, float eps, int silu,
                                    mvd_stream_t stream) {
  MVD_CHECK_ARG(a && b && y_sp && gamma && beta && B > 0 && hw > 0, "mvd_concat_groupnorm: null pointer / bad shape");
  MVD_CHECK_ARG((Ca + Cb) % 32 == 0 && mvd_concat_groupnorm_fits(Ca, Cb, hw, groups),
                "mvd_concat_groupnorm: needs (Ca+Cb) %% 32 == 0, even Ca / Cb / group width <= 128 and hw * (C / groups) * 4 <= 128 KiB of LDS "
                "(Ca=%d Cb=%d hw=%d groups=%d): use mvd_concat_channels + mvd_groupnorm_from_stats", Ca, Cb, hw, groups);
  MVD_CHECK_ARG(((uintptr_t)a & 7) == 0 && ((uintptr_t)b & 7) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)y_sp & 127) == 0 &&
                    ((uintptr_t)raw_sp & 127) == 0, "mvd_concat_groupnorm: alignment");
  const size_t lds = (size_t)hw * ((Ca + Cb) / groups) * 4;
  {                   // (more than the default 64 KiB of dynamic LDS: the largest groups of a step are 1024 rows x 30 channels)
    static unsigned long long raised = 0;
    const hipError_t e = mvd_raise_dynamic_lds((const void*)concat_gn_kernel, 128 * 1024, &raised);
    MVD_CHECK_ARG(e == hipSuccess, "mvd_concat_groupnorm: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(concat_gn_kernel, dim3(B * groups), dim3(MVD_CGN_THREADS), lds, (hipStream_t)stream, a, Ca, b, Cb, out, (u16*)raw_sp,
                     (u16*)y_sp, gamma, beta, gn_stats, hw, groups, eps, silu);
  MVD_CHECK_LAUNCH("mvd_concat_groupnorm");
  return 0;
}

extern "C" int mvd_area_pool(const float* vol, void* out_sp, int B, int S, int D, int C, int factor, int ldp,
                             mvd_stream_t stream) {
  MVD_CHECK_ARG(vol && out_sp && C % 32 == 0 && B > 0 && S > 0 && D > 0 && C % 4 == 0 && factor >= 1 && S % factor == 0,
                "mvd_area_pool: bad arguments");
  if (ldp == 0) ldp = C;
  MVD_CHECK_ARG(ldp >= C && ldp % 32 == 0 && ((uintptr_t)out_sp & 127) == 0, "mvd_area_pool: ldp=%d must be >= C, %% 32 == 0", ldp);
  const int So = S / factor;
  const size_t total = (size_t)B * So * So * D * (C / 4);
  hipLaunchKernelGGL(area_pool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)vol,
                     (u16*)out_sp, B, S, D, C / 4, factor, ldp);
  MVD_CHECK_LAUNCH("mvd_area_pool");
  return 0;
}

extern "C" int mvd_fill_zero(float* p, size_t n, mvd_stream_t stream) {
  MVD_CHECK_ARG(p != nullptr, "mvd_fill_zero: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n);
  MVD_CHECK_LAUNCH("mvd_fill_zero");
  return 0;
}

extern "C" int mvd_timestep_embedding(const float* steps, const int* iter, const float* freqs, float* out, int dim,
                                      mvd_stream_t stream) {
  MVD_CHECK_ARG(steps && iter && freqs && out && dim > 0 && dim % 2 == 0, "mvd_timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, steps, iter, freqs, out, dim);
  MVD_CHECK_LAUNCH("mvd_timestep_embedding");
  return 0;
}

extern "C" int mvd_advance_iter(int* iter, mvd_stream_t stream) {
  MVD_CHECK_ARG(iter != nullptr, "mvd_advance_iter: null pointer");
  hipLaunchKernelGGL(advance_iter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, iter);
  MVD_CHECK_LAUNCH("mvd_advance_iter");
  return 0;
}

extern "C" int mvd_cfg_ddim_update(const float* eps_nhwc, int ldc, float* x, float* x0, float* eps_out,
                                   const float* ddim_noise, size_t noise_stride, const float* steps, const int* iter, int V,
                                   int S, int cfg, float cfg_scale, int do_update, mvd_stream_t stream) {
  MVD_CHECK_ARG(eps_nhwc && steps && iter && V > 0 && S > 0 && ldc >= 5, "mvd_cfg_ddim_update: bad arguments");
  if (do_update) MVD_CHECK_ARG(x && x0 && ddim_noise, "mvd_cfg_ddim_update: update needs x, x0, ddim_noise");
  const size_t total = (size_t)V * 5 * S * S;
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, eps_nhwc, ldc, x, x0, eps_out,
                     ddim_noise, steps, iter, V, S, cfg, cfg_scale, do_update, noise_stride);
  MVD_CHECK_LAUNCH("mvd_cfg_ddim_update");
  return 0;
}
