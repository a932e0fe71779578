// GEMM / implicit-GEMM 3x3 convolution on bf16 MFMA (gfx950): both operands arrive as split-bf16 planes and are
// DMA'd straight into LDS; fused epilogues.  See include/mvd_hip.h (mvd_gemm) for the contract.
//
// Operands
//   A : activations as two bf16 planes (hi, lo) with x ~= hi + lo, written by the PRODUCING kernel (norms, attention,
//       previous GEMM epilogue, ...) -- same bytes as fp32, no conversion work inside the GEMM.
//   B : weights packed once at load time into 1 KiB micro-tiles [K/32][N/16][hi,lo][16 n][32 k].
// Structure (per workgroup): block tile BM x BN, BK = 32, WM x WN waves, each wave a (BM/WM) x (BN/WN) sub-tile of
// 16x16x32 MFMAs.  A k-tile of both operands is a set of 1 KiB granules (16 rows x 64 B); each wave instruction of
// `global_load_lds_dwordx4` moves one granule global -> LDS with no VGPR round trip (LDS destination is lane-linear,
// so the bank-conflict swizzle is applied to the per-lane SOURCE address and to the fragment reads: 16-byte chunk c of
// row r lives at slot r*4 + (c ^ ((-(r>>2)) & 3)), which is conflict-free for the 16-lane ds_read_b128 groups).
// Two LDS stages: the DMA of k-tile t+1 is in flight while the MFMAs of k-tile t run; rows/columns outside the
// problem (M/N edges, conv zero padding) source a 16-byte zero page.
// NS = 1: acc += A_hi*B_hi.   NS = 3: acc += A_lo*B_hi + A_hi*B_lo + A_hi*B_hi   (fp32 accumulate).
#include <stdlib.h>

#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4];

struct GemmParams {
  mvd_gemm_desc d;
  int nk;        // K / 32
  int nt16;      // packed N / 16
  int kt_per_split;
  int splits;
};

// ------------------------------------------------------------------------------------------------ epilogue
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == MVD_ACT_GELU) return gelu_erf(v);
  if (act == MVD_ACT_SILU) return silu_f(v);
  return v;
}

__device__ __forceinline__ void store_out(const mvd_gemm_desc& d, int m, int n, float v) {
  if (d.out) d.out[(size_t)m * d.ldo + n] = v;
  if (d.out_hi) {
    u16 hi, lo;
    split_bf16(v, hi, lo);
    ((u16*)d.out_hi)[(size_t)m * d.ldp + n] = hi;
    ((u16*)d.out_lo)[(size_t)m * d.ldp + n] = lo;
  }
}

__device__ __forceinline__ void epi_store_elem(const mvd_gemm_desc& d, int m, int n, float v) {
  if (d.epi == MVD_EPI_STORE && n >= d.n_store) return;  // padded columns (bias / res have n_store entries)
  if (d.bias) v += d.bias[n];
  if (d.bias_b) v += d.bias_b[(size_t)(m / d.rows_per_batch) * d.N + n];
  if (d.epi == MVD_EPI_QKV) {
    const int C = d.heads * d.dhead;
    const int which = n / C;
    const int cc = n - which * C;
    const int head = cc / d.dhead;
    const int dd = cc - head * d.dhead;
    const int b = m / d.L;
    const int tok = m - b * d.L;
    if (which == 0) v *= d.qscale;
    u16 hi, lo;
    split_bf16(v, hi, lo);
    if (which < 2) {
      const int dq = (d.dhead + 31) & ~31;
      const size_t idx = ((size_t)(b * d.heads + head) * d.Lpad + tok) * dq + dd;
      u16* ph = (u16*)(which == 0 ? d.q_hi : d.k_hi);
      u16* pl = (u16*)(which == 0 ? d.q_lo : d.k_lo);
      ph[idx] = hi;
      pl[idx] = lo;
    } else {
      const int dv = (d.dhead + 15) & ~15;
      const size_t idx = ((size_t)(b * d.heads + head) * dv + dd) * d.Lpad + tok;
      ((u16*)d.vt_hi)[idx] = hi;
      ((u16*)d.vt_lo)[idx] = lo;
    }
    return;
  }
  v = apply_act(v, d.act);
  if (d.colscale) v *= d.colscale[n];
  if (d.res) v += d.res[(size_t)m * d.ldr + n];
  store_out(d, m, n, v);
}

// value / gate pair -> one output column (packed column p: block of 32 = 16 value + 16 gate)
__device__ __forceinline__ void epi_geglu_elem(const mvd_gemm_desc& d, int m, int p_value, float v, float g) {
  const int col = (p_value >> 5) * 16 + (p_value & 15);
  const int half = d.N >> 1;
  if (d.bias) {
    v += d.bias[col];
    g += d.bias[half + col];
  }
  store_out(d, m, col, v * gelu_erf(g));
}

// ------------------------------------------------------------------------------------------------ main kernel
template <int BM, int BN, int WM, int WN, int NS, int AMODE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int NPL = (NS == 3) ? 2 : 1;
  constexpr int A_GRAN = (BM / 16) * NPL, B_GRAN = (BN / 16) * NPL;
  constexpr int STAGE = (A_GRAN + B_GRAN) * 1024;
  constexpr int AI = A_GRAN / NW, BI = B_GRAN / NW;   // granules per wave per k-tile
  static_assert(A_GRAN % NW == 0 && B_GRAN % NW == 0, "granules must divide evenly over the waves");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const mvd_gemm_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int kt0 = blockIdx.z * p.kt_per_split;
  const int kt1 = min(p.nk, kt0 + p.kt_per_split);

  // ---- per-lane staging roles.  Lane l of a granule fills slot l: row r = l>>2, stored chunk (l&3) holds source
  //      chunk c = (l&3) ^ f(r),  f(r) = (-(r>>2)) & 3.
  const int gr = lane >> 2;
  const int gc = (lane & 3) ^ ((-(gr >> 2)) & 3);
  const u16* zero = (const u16*)g_zero_page;

  const u16* a_src[AI];     // per A granule: source row base (k = 0) or zero page
  int a_oy[AI], a_ox[AI];
  bool a_ok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int gi = wave + i * NW;            // A granule index: mt * NPL + plane
    const int mt = gi / NPL, plane = gi % NPL;
    const int m = m0 + mt * 16 + gr;
    a_ok[i] = m < d.M;
    const u16* base = (const u16*)(plane ? d.A_lo : d.A_hi);
    if (AMODE == MVD_A_DENSE) {
      a_src[i] = base + (size_t)(a_ok[i] ? m : 0) * d.lda + gc * 8;
      a_oy[i] = a_ox[i] = 0;
    } else {
      const int hw = d.Hout * d.Wout;
      const int mm = a_ok[i] ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      a_oy[i] = rem / d.Wout;
      a_ox[i] = rem - a_oy[i] * d.Wout;
      a_src[i] = base + (size_t)b * d.Hin * d.Win * d.Cin + gc * 8;
    }
  }
  const u16* b_src[BI];     // per B granule: micro-tile base at kt = 0 (+ this lane's chunk) or null (-> zero page)
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int gi = wave + i * NW;            // B granule index: nt * NPL + plane
    const int nt = (n0 >> 4) + gi / NPL, plane = gi % NPL;
    b_src[i] = nt < p.nt16 ? (const u16*)d.Wp + ((size_t)nt * 2 + plane) * 512 + gr * 32 + gc * 8 : nullptr;
  }
  const size_t b_kstride = (size_t)p.nt16 * 1024;   // elements between consecutive k-tiles of the packed weight

  auto stage = [&](int buf, int kt) {
    unsigned char* sbase = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const u16* src;
      if (AMODE == MVD_A_DENSE) {
        src = a_ok[i] ? a_src[i] + kt * 32 : zero;
      } else {
        const int k0 = kt * 32;
        const int tap = k0 / d.Cin;
        const int c0 = k0 - tap * d.Cin;
        const int ky = tap / 3, kx = tap - ky * 3;
        int iy, ix;
        bool ok = a_ok[i];
        if (d.upsample) {
          const int uy = a_oy[i] + ky - 1, ux = a_ox[i] + kx - 1;
          ok = ok && uy >= 0 && uy < d.Hout && ux >= 0 && ux < d.Wout;
          iy = uy >> 1;
          ix = ux >> 1;
        } else {
          iy = a_oy[i] * d.stride + ky - 1;
          ix = a_ox[i] * d.stride + kx - 1;
          ok = ok && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win;
        }
        src = ok ? a_src[i] + ((size_t)iy * d.Win + ix) * d.Cin + c0 : zero;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sbase + (wave + i * NW) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const u16* src = b_src[i] ? b_src[i] + (size_t)kt * b_kstride : zero;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(sbase + (A_GRAN + wave + i * NW) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offset inside a granule: row = lane&15, k-chunk = lane>>4 (swizzled)
  const int frow = lane & 15;
  const int foff = (frow * 4 + ((lane >> 4) ^ ((-(frow >> 2)) & 3))) * 16;

  if (kt0 < kt1) stage(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
    const unsigned char* sA = smem + cur * STAGE + foff;
    const unsigned char* sB = sA + A_GRAN * 1024;
    bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *(const bf16x8*)(sA + ((wm * TM + i) * NPL) * 1024);
      if (NS == 3) al[i] = *(const bf16x8*)(sA + ((wm * TM + i) * NPL + 1) * 1024);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *(const bf16x8*)(sB + ((wn * TN + j) * NPL) * 1024);
      if (NS == 3) bl[j] = *(const bf16x8*)(sB + ((wn * TN + j) * NPL + 1) * 1024);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (NS == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue.  C layout: row = (lane>>4)*4 + r, col = lane&15.
  const int crow = (lane >> 4) * 4;
  const int ccol = lane & 15;
  const int wm0 = m0 + wm * (BM / WM), wn0 = n0 + wn * (BN / WN);
  if (p.splits > 1) {
    float* ws = d.workspace + (size_t)blockIdx.z * d.M * d.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = wn0 + j * 16 + ccol;
        if (n >= d.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = wm0 + i * 16 + crow + r;
          if (m < d.M) ws[(size_t)m * d.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  if (d.epi == MVD_EPI_GEGLU) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const int n = wn0 + j * 16 + ccol;
        if (n >= d.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = wm0 + i * 16 + crow + r;
          if (m < d.M) epi_geglu_elem(d, m, n, acc[i][j][r], acc[i][j + 1][r]);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = wn0 + j * 16 + ccol;
      if (n >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = wm0 + i * 16 + crow + r;
        if (m < d.M) epi_store_elem(d, m, n, acc[i][j][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------ split-K reduce
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const mvd_gemm_desc& d = p.d;
  const size_t MN = (size_t)d.M * d.N;
  if (d.epi == MVD_EPI_GEGLU) {
    const size_t total = (size_t)d.M * (d.N >> 1);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
      const int m = (int)(e / (d.N >> 1));
      const int col = (int)(e - (size_t)m * (d.N >> 1));
      const int pv = (col >> 4) * 32 + (col & 15);
      float v = 0.f, g = 0.f;
      for (int z = 0; z < p.splits; ++z) {
        v += d.workspace[z * MN + (size_t)m * d.N + pv];
        g += d.workspace[z * MN + (size_t)m * d.N + pv + 16];
      }
      epi_geglu_elem(d, m, pv, v, g);
    }
    return;
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < MN; e += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < p.splits; ++z) v += d.workspace[z * MN + e];
    const int m = (int)(e / d.N);
    const int n = (int)(e - (size_t)m * d.N);
    epi_store_elem(d, m, n, v);
  }
}

template <int BM, int BN, int WM, int WN>
void launch_cfg(const GemmParams& p, hipStream_t s) {
  dim3 grid(cdiv(p.d.N, BN), cdiv(p.d.M, BM), p.splits), block(WM * WN * 64);
  const bool conv = p.d.a_mode == MVD_A_CONV3X3;
  const bool x3 = p.d.prec == MVD_PREC_BF16X3;
  if (!conv && x3) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 3, MVD_A_DENSE>), grid, block, 0, s, p);
  if (!conv && !x3) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, MVD_A_DENSE>), grid, block, 0, s, p);
  if (conv && x3) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 3, MVD_A_CONV3X3>), grid, block, 0, s, p);
  if (conv && !x3) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, MVD_A_CONV3X3>), grid, block, 0, s, p);
}

}  // namespace

extern "C" int mvd_gemm(const mvd_gemm_desc* dp, mvd_stream_t stream) {
  MVD_CHECK_ARG(dp != nullptr, "mvd_gemm: null descriptor");
  GemmParams p;
  p.d = *dp;
  mvd_gemm_desc& d = p.d;
  MVD_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "mvd_gemm: bad sizes M=%d N=%d K=%d", d.M, d.N, d.K);
  MVD_CHECK_ARG(d.K % 32 == 0, "mvd_gemm: K=%d must be a multiple of 32 (pad the packed weight)", d.K);
  MVD_CHECK_ARG(d.N % 16 == 0, "mvd_gemm: N=%d must be a multiple of 16 (pad the packed weight)", d.N);
  MVD_CHECK_ARG(d.prec == MVD_PREC_BF16 || d.prec == MVD_PREC_BF16X3, "mvd_gemm: bad prec %d", d.prec);
  MVD_CHECK_ARG(d.A_hi && d.Wp && (d.A_lo || d.prec == MVD_PREC_BF16), "mvd_gemm: null operand");
  MVD_CHECK_ARG(((uintptr_t)d.A_hi & 15) == 0 && ((uintptr_t)d.A_lo & 15) == 0 && ((uintptr_t)d.Wp & 15) == 0,
                "mvd_gemm: operands must be 16-byte aligned");
  if (d.a_mode == MVD_A_CONV3X3) {
    MVD_CHECK_ARG(d.Cin % 32 == 0 && d.K == 9 * d.Cin, "mvd_gemm: conv needs Cin %% 32 == 0 and K == 9*Cin (Cin=%d K=%d)", d.Cin, d.K);
    MVD_CHECK_ARG(d.M == d.B * d.Hout * d.Wout, "mvd_gemm: conv M mismatch");
    MVD_CHECK_ARG(d.stride == 1 || d.stride == 2, "mvd_gemm: conv stride must be 1 or 2");
    if (d.upsample) MVD_CHECK_ARG(d.stride == 1 && d.Hout == 2 * d.Hin && d.Wout == 2 * d.Win, "mvd_gemm: upsample geometry");
  } else {
    MVD_CHECK_ARG(d.a_mode == MVD_A_DENSE, "mvd_gemm: bad a_mode");
    MVD_CHECK_ARG(d.lda >= d.K && d.lda % 8 == 0, "mvd_gemm: lda=%d must be >= K=%d and a multiple of 8", d.lda, d.K);
  }
  if (d.out_hi || d.out_lo) MVD_CHECK_ARG(d.out_hi && d.out_lo && d.ldp > 0, "mvd_gemm: plane output needs both planes and ldp");
  if (d.epi == MVD_EPI_STORE) {
    MVD_CHECK_ARG(d.out != nullptr || d.out_hi != nullptr, "mvd_gemm: no output");
    if (d.n_store <= 0 || d.n_store > d.N) d.n_store = d.N;
    if (d.bias_b) MVD_CHECK_ARG(d.rows_per_batch > 0, "mvd_gemm: bias_b needs rows_per_batch");
  } else if (d.epi == MVD_EPI_GEGLU) {
    MVD_CHECK_ARG((d.out != nullptr || d.out_hi != nullptr) && d.N % 32 == 0, "mvd_gemm: GEGLU needs an output and N %% 32 == 0");
    d.bias_b = nullptr;
  } else if (d.epi == MVD_EPI_QKV) {
    MVD_CHECK_ARG(d.q_hi && d.q_lo && d.k_hi && d.k_lo && d.vt_hi && d.vt_lo, "mvd_gemm: QKV planes missing");
    MVD_CHECK_ARG(d.heads > 0 && d.dhead > 0 && d.N == 3 * d.heads * d.dhead, "mvd_gemm: QKV needs N == 3*heads*dhead");
    MVD_CHECK_ARG(d.L > 0 && d.M % d.L == 0 && d.Lpad >= d.L, "mvd_gemm: QKV needs M %% L == 0");
  } else {
    MVD_CHECK_ARG(false, "mvd_gemm: bad epilogue %d", d.epi);
  }
  p.nk = d.K / 32;
  p.nt16 = d.N / 16;
  // tile selection: 128x128 (8 waves, 2 workgroups / CU) once the grid fills the chip, else 64x64 (4 waves)
  const long tiles128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
  bool big = tiles128 >= 256;
  if (const char* e = getenv("MVD_GEMM_TILE")) big = atoi(e) >= 128;
  const int BM = big ? 128 : 64, BN = big ? 128 : 64;
  int splits = d.splitk;
  const long tiles = (long)cdiv(d.M, BM) * cdiv(d.N, BN);
  if (splits == 0) {  // auto: fill the chip on the small-M, huge-K (weight-bandwidth-bound) layers
    splits = 1;
    if (tiles < 256 && p.nk >= 32) {
      splits = (int)((768 + tiles - 1) / tiles);
      if (splits > p.nk / 8) splits = p.nk / 8;
    }
  }
  if (splits < 1) splits = 1;
  if (splits > p.nk) splits = p.nk;
  if (splits > 1) {
    if (d.workspace == nullptr) splits = 1;
    else {
      const size_t cap = d.workspace_elems / ((size_t)d.M * d.N);
      if ((size_t)splits > cap) splits = cap < 1 ? 1 : (int)cap;
    }
  }
  p.kt_per_split = cdiv(p.nk, splits);
  p.splits = cdiv(p.nk, p.kt_per_split);
  hipStream_t s = (hipStream_t)stream;
  if (big)
    launch_cfg<128, 128, 2, 4>(p, s);
  else
    launch_cfg<64, 64, 2, 2>(p, s);
  MVD_CHECK_LAUNCH("mvd_gemm");
  if (p.splits > 1) {
    const size_t total = (size_t)d.M * d.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
    MVD_CHECK_LAUNCH("mvd_gemm/splitk_reduce");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ packing
namespace {
// one thread per packed element pair (hi, lo)
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, u16* __restrict__ out, int N, int K,
                                                   int Np, int Kp, int ldw, int geglu, int conv_cin, int conv_cin_pad) {
  const size_t total = (size_t)Np * Kp;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int n = (int)(e / Kp);   // packed row
    const int k = (int)(e - (size_t)n * Kp);
    int src_n = n;
    if (geglu) {
      const int blk = n >> 5, within = n & 31;
      src_n = within < 16 ? blk * 16 + within : (N >> 1) + blk * 16 + (within - 16);
    }
    float v = 0.f;
    if (src_n < N) {
      if (conv_cin > 0) {
        const int tap = k / conv_cin_pad;
        const int ci = k - tap * conv_cin_pad;
        if (ci < conv_cin && tap < 9) v = w[((size_t)src_n * conv_cin + ci) * 9 + tap];
      } else if (k < K) {
        v = w[(size_t)src_n * ldw + k];
      }
    }
    u16 hi, lo;
    split_bf16(v, hi, lo);
    const int kt = k >> 5, kk = k & 31, nt = n >> 4, nn = n & 15;
    const size_t base = ((size_t)kt * (Np >> 4) + nt) * 1024 + nn * 32 + kk;
    out[base] = hi;
    out[base + 512] = lo;
  }
}

// fp32 (rows, cols) with leading dim ldx -> bf16 planes (rows, ldp) ; columns [cols, ldp) are zero filled
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, u16* __restrict__ hi, u16* __restrict__ lo,
                                                           size_t rows, int cols, int ldx, int ldp) {
  const int c4 = ldp >> 2;
  const size_t total = rows * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / c4;
    const int c = (int)(e - r * c4) * 4;
    u16 h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = (c + j) < cols ? x[r * ldx + c + j] : 0.f;
      split_bf16(v, h[j], l[j]);
    }
    *(uint2*)(hi + r * ldp + c) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *(uint2*)(lo + r * ldp + c) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  }
}
}  // namespace

extern "C" size_t mvd_packed_weight_bytes(int N, int K) {
  const size_t Np = (size_t)((N + 15) & ~15), Kp = (size_t)((K + 31) & ~31);
  return Np * Kp * 4;
}

extern "C" int mvd_pack_linear_weight(const float* w, int N, int K, int ldw, int geglu, void* packed, mvd_stream_t stream) {
  MVD_CHECK_ARG(w && packed && N > 0 && K > 0 && ldw >= K, "mvd_pack_linear_weight: bad arguments");
  if (geglu) MVD_CHECK_ARG(N % 32 == 0, "mvd_pack_linear_weight: geglu needs N %% 32 == 0");
  const int Np = (N + 15) & ~15, Kp = (K + 31) & ~31;
  const size_t total = (size_t)Np * Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (u16*)packed, N, K, Np, Kp, ldw,
                     geglu, 0, 0);
  MVD_CHECK_LAUNCH("mvd_pack_linear_weight");
  return 0;
}

extern "C" int mvd_pack_conv3x3_weight(const float* w, int Cout, int Cin, int cin_pad, void* packed, mvd_stream_t stream) {
  MVD_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && cin_pad >= Cin && cin_pad % 32 == 0,
                "mvd_pack_conv3x3_weight: bad arguments (cin_pad must be a multiple of 32)");
  const int Np = (Cout + 15) & ~15, Kp = 9 * cin_pad;
  const size_t total = (size_t)Np * Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (u16*)packed, Cout, Kp, Np, Kp, 0,
                     0, Cin, cin_pad);
  MVD_CHECK_LAUNCH("mvd_pack_conv3x3_weight");
  return 0;
}

extern "C" int mvd_split_planes(const float* x, void* hi, void* lo, size_t rows, int cols, int ldx, int ldp,
                                mvd_stream_t stream) {
  MVD_CHECK_ARG(x && hi && lo && rows > 0 && cols > 0 && ldx >= cols && ldp >= cols && ldp % 8 == 0,
                "mvd_split_planes: bad arguments (ldp must be a multiple of 8)");
  const size_t total = rows * (size_t)(ldp / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (u16*)hi, (u16*)lo, rows, cols,
                     ldx, ldp);
  MVD_CHECK_LAUNCH("mvd_split_planes");
  return 0;
}
