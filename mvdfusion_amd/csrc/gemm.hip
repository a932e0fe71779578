// GEMM / implicit-GEMM 3x3 convolution on 16-bit MFMA (gfx950): both operands arrive as split planes (hi + lo in the
// MFMA operand type, fp16 by default) and are DMA'd straight into LDS; fused epilogues.  See include/mvd_hip.h (mvd_gemm)
// for the contract.
//
// Operands
//   A : activations in the "split planes" format (common.hpp): x ~= hi + lo, per row and 32-element k-block
//       [32 hi | 32 lo] = one 128-byte line, written by the PRODUCING kernel (norms, attention, previous GEMM
//       epilogue, ...) -- same bytes as fp32, no conversion work inside the GEMM.  Conv: the NHWC image rows; the K order
//       is (32-channel block, tap, channel) so the nine taps of a pixel line are consecutive k-tiles (L2 hits), and a
//       per-workgroup LDS table holds the source offset of every (tile row, tap).
//   B : weights packed once at load time into 2 KiB micro-tiles [K/32][N/16][16 n][32 hi | 32 lo], pre-scaled by a power
//       of two (acc_scale undoes it).
// Structure (per workgroup): block tile BM x BN, BK = 32, WM x WN waves, each wave a (BM/WM) x (BN/WN) sub-tile of 16x16x32
// MFMAs.  Tiles: 64x64 (4 waves), 128x128 (8 waves), and the 80-column family 128x80 / 64x80 (4 waves) and 128x160 (8 waves)
// for the N = 320 * k layers of the UNet: no N padding (320 = 4 x 80), exactly 256 workgroups for M = 8192, N = 320, and
// fewer L2->LDS bytes per MFMA than 64x64 -- the kernel is bound by operand delivery (~25 B/clk/CU of LDS-DMA), so the tile
// is chosen for bytes per MFMA and for how evenly the grid fills the 256 CUs.  A k-tile of both operands is a set of 1 KiB granules (8 rows x one
// full 128-byte line each); every wave instruction of `global_load_lds_dwordx4` moves one granule global -> LDS with no
// VGPR round trip (the LDS destination is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
// address and to the fragment reads: 16-byte chunk cc (0-3 hi, 4-7 lo) of row r of a 16-row block lives at slot
// (r&7)*8 + (cc ^ (r>>1)), which is conflict-free for the 16-lane ds_read_b128 groups).  Running source pointers: the
// k loop carries no address arithmetic beyond one add per granule; rows/columns outside the problem (M/N edges, conv
// zero padding) source a 16-byte zero page.
// Loop variants (template STAGES).  4 = STAGGERED (8-wave tiles only, three LDS buffers): the two wavefronts that share a SIMD
// run half an iteration apart -- in every phase one of them issues its LDS-DMA share of k-tile t+2 and reads its fragments of
// k-tile t (memory phase) while its partner runs the MFMAs of its own current k-tile, one raw s_barrier per phase.  The DMA /
// ds_read issue time (60-185 cycles per 1 KiB DMA instruction) that otherwise sits between two MFMA bursts of a SIMD is then
// covered by the partner's MFMAs.  Two LDS buffers for the other two variants: 2 = plain (DMA of k-tile t+1 in flight under the MFMAs of t;
// two co-resident workgroups per CU hide each other's waits), 3 = register-pipelined (fragments of t+1 read and DMA of
// t+2 issued under the MFMAs of t).  One `s_waitcnt vmcnt lgkmcnt` + raw `s_barrier` per k-tile.
// NS = 1: acc += A_hi*B_hi.  NS = 3 (the default, "f16x3"): acc += A_lo*B_hi + A_hi*B_lo + A_hi*B_hi.  NS = 4: + A_lo*B_lo first.  The
// fourth product is NOT free: measured -5 ... -7 % step time for x3 (the long-K convolutions are ~50 % MFMA-bound; the short-K
// projections do not care), and its 2^-22 term is below the fp32 accumulation noise -- indistinguishable in the 50-step trajectory
// (DESIGN.md section 4).
// Also here: the split-K reduce kernels, among them splitk_gn_kernel -- reduce + epilogue + GroupNorm / SiLU of the output (optionally over
// its concatenation with a skip tensor) in one launch, a workgroup per (image, group) with the group's values in LDS.
#include <stdlib.h>
#include <type_traits>

#include "gemm_common.hpp"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4];

// ------------------------------------------------------------------------------------------------ epilogue
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == MVD_ACT_GELU) return gelu_erf(v);
  if (act == MVD_ACT_SILU) return silu_f(v);
  if (act == MVD_ACT_QUICKGELU) return v / (1.0f + expf(-1.702f * v));       // x * sigmoid(1.702 x) (OpenAI CLIP QuickGELU)
  return v;
}

__device__ __forceinline__ void store_out(const mvd_gemm_desc& d, int m, int n, float v) {
  if (d.out) d.out[(size_t)m * d.ldo + n] = v;
  if (d.out_sp) store_sp1((u16*)d.out_sp, (size_t)m, d.ldp, n, v);
}

// scalar element path (split-K reduce kernel, ragged n_store edge)
__device__ __forceinline__ float epi_store_elem(const mvd_gemm_desc& d, int m, int n, float v) {
  if (d.epi == MVD_EPI_STORE && n >= d.n_store) return 0.f;  // padded columns (bias / res have n_store entries)
  v *= d.acc_scale;
  if (d.bias) v += d.bias[n];
  if (d.bias_b) v += d.bias_b[(size_t)(m / d.rows_per_batch) * d.ldbb + n];
  if (d.epi == MVD_EPI_QKV) {
    const int C = d.heads * d.dhead;
    const int which = n / C;
    const int cc = n - which * C;
    const int head = cc / d.dhead;
    const int dd = cc - head * d.dhead;
    const int b = m / d.L;
    const int tok = m - b * d.L;
    if (which == 0) v *= d.qscale;
    if (which < 2) {
      const int dq = mvd_attn_dpad(d.dhead);
      const size_t idx = ((size_t)(b * d.heads + head) * d.Lpad + tok) * dq + dd;
      store_planes1((u16*)(which == 0 ? d.q_hi : d.k_hi), (u16*)(which == 0 ? d.q_lo : d.k_lo), idx, v);
    } else {
      const int dv = (d.dhead + 15) & ~15;
      const size_t idx = ((size_t)(b * d.heads + head) * dv + dd) * d.Lpad + tok;
      store_planes1((u16*)d.vt_hi, (u16*)d.vt_lo, idx, v);
    }
    return v;
  }
  v = apply_act(v, d.act);
  if (d.colscale) v *= d.colscale[n];
  if (d.res) v += d.res[(size_t)m * d.ldr + n];
  store_out(d, m, n, v);
  return v;
}

// value / gate pair -> one output column (packed column p: block of 32 = 16 value + 16 gate)
__device__ __forceinline__ void epi_geglu_elem(const mvd_gemm_desc& d, int m, int p_value, float v, float g) {
  const int col = (p_value >> 5) * 16 + (p_value & 15);
  const int half = d.N >> 1;
  v *= d.acc_scale;
  g *= d.acc_scale;
  if (d.bias) {
    v += d.bias[col];
    g += d.bias[half + col];
  }
  store_out(d, m, col, v * gelu_erf(g));
}

// four consecutive columns n..n+3 of row m (all inside N): coalesced 16-byte traffic; returns the final values
// (epi_value4: operands + arithmetic, epi_put4: the stores -- callers with several rows per thread run all the values before the first store:
//  a load behind a conditional store waits for its acknowledgement)
__device__ __forceinline__ float4 epi_value4(const mvd_gemm_desc& d, int m, int n, float4 v) {
  v.x *= d.acc_scale; v.y *= d.acc_scale; v.z *= d.acc_scale; v.w *= d.acc_scale;
  if (d.bias) {
    const float4 b = *(const float4*)(d.bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (d.bias_b) {
    const float4 b = *(const float4*)(d.bias_b + (size_t)(m / d.rows_per_batch) * d.ldbb + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (d.act) {
    v.x = apply_act(v.x, d.act); v.y = apply_act(v.y, d.act); v.z = apply_act(v.z, d.act); v.w = apply_act(v.w, d.act);
  }
  if (d.colscale) {
    const float4 g = *(const float4*)(d.colscale + n);
    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
  }
  if (d.res) {
    const float4 r = *(const float4*)(d.res + (size_t)m * d.ldr + n);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  return v;
}
__device__ __forceinline__ void epi_put4(const mvd_gemm_desc& d, int m, int n, const float4& v) {
  if (d.out) *(float4*)(d.out + (size_t)m * d.ldo + n) = v;
  if (d.out_sp) store_sp4((u16*)d.out_sp, (size_t)m, d.ldp, n, v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 epi_store4(const mvd_gemm_desc& d, int m, int n, float4 v) {
  v = epi_value4(d, m, n, v);
  epi_put4(d, m, n, v);
  return v;
}

// conv_patch_kernel: slots of the input patch (128 B each) a workgroup may hold (BM = 128: 288 = 8 images of 4x4 with halo)
#define MVD_PATCH_SLOTS_MAX 288
// ... and the depth of its ring of weight stages, by tile width (two patch buffers of 37 KiB + the ring fit the CU's 160 KiB)
__host__ __device__ constexpr int conv_patch_ring(int bn, int waves) {
  const int bstage = ((bn / 8 + waves - 1) / waves) * waves;       // KiB
  const int fit = (160 - 2 * (MVD_PATCH_SLOTS_MAX / 8 + 1) - 2) / bstage;
  return fit > 4 ? 4 : fit;
}

// LayerNorm folded into a GEMM (mvd_gemm_desc.ln_stats): mean and 1/std of row m of the A operand from the producer's per-slot
// {sum, sum of squares} partials -- summed in slot order in double (deterministic; var = E[x^2] - mean^2 needs the headroom).
__device__ __forceinline__ float2 ln_row_stats(const mvd_gemm_desc& d, int m) {
  const int cnt = d.ln_count[0];
  const float2* p = (const float2*)d.ln_stats + (size_t)m * d.ln_ld;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < cnt; i += 4) {          // four independent loads in flight per round trip; added in slot order
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = p[min(i + j, cnt - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i + j < cnt) {
        s += (double)v[j].x;
        q += (double)v[j].y;
      }
    }
  }
  const double mean = s / (double)d.ln_dim;
  double var = q / (double)d.ln_dim - mean * mean;
  if (var < 0.0) var = 0.0;
  return make_float2((float)mean, (float)(1.0 / sqrt(var + (double)d.ln_eps)));
}

#ifdef MVD_STAMP
// profiling build (tools/probes/stamp.sh): cycle stamps of workgroups 0 and 100, consumer waves 0..3, into d.workspace (int64[2][4][16])
#define MVD_STAMP_AT(d, wave, slot)                                                                                       \
  do {                                                                                                                    \
    if ((blockIdx.x == 0 || blockIdx.x == 100) && (wave) < 4 && (threadIdx.x & 63) == 0)                                  \
      ((long long*)(d).workspace)[(blockIdx.x == 100 ? 64 : 0) + (wave) * 16 + (slot)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define MVD_STAMP_AT(d, wave, slot) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------ tile epilogue
// Shared by gemm_kernel and conv_patch_kernel: the wave's accumulator tile is transposed through LDS (the stage buffers are free: the
// caller has passed a workgroup barrier after its last fragment read) so that global traffic is row-contiguous 16-byte accesses.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], unsigned char* smem, int m0,
                                              int n0, int lane, int wave, const float* s_rows = nullptr) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int LDW = WTN + 4;                        // fp32 pitch of the epilogue staging tile
  constexpr int C4 = WTN / 4;                         // float4 columns of a wave tile row
  const mvd_gemm_desc& d = p.d;
  const int wm = wave / WN, wn = wave % WN;
  //      (the final barrier above guarantees nobody still reads the stage buffers; each wave owns a private region)
  float* sC = (float*)smem + wave * (WTM * LDW);
  MVD_STAMP_AT(d, wave, 4);
  {
    const int crow = (lane >> 4) * 4, ccol = lane & 15;   // C layout: row = (lane>>4)*4 + r, col = lane&15
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(i * 16 + crow + r) * LDW + j * 16 + ccol] = acc[i][j][r];
  }
  const int wm0 = m0 + wm * WTM, wn0 = n0 + wn * WTN;
#ifdef MVD_STAMP
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  MVD_STAMP_AT(d, wave, 5);
  if (wn0 >= d.N) return;
  // LayerNorm of the A rows folded in: y = rstd (acc - mean colsum) + bias; {mean, rstd} of the block tile's rows were gathered into LDS
  // by the kernel's prologue (gemm_kernel: ln_gather_rows)
  const bool lnf = s_rows != nullptr;
  const float* sR = s_rows + (wave / WN) * WTM * 2;
  if (p.splits > 1) {   // raw partial sums -> workspace slab; splitk_reduce_kernel sums the slabs and applies the epilogue.
    // (Reducing inside this kernel -- last-arriving workgroup per tile behind an agent-scope release/acquire -- was
    //  built and measured: bit-identical, but 15 % slower per step.  A 128x128 tile has 64 KB slabs, far above the
    //  few tens of KB where that hand-off pays, and its cache-wide write-back / invalidate disturbs the operand
    //  streams of the other workgroups.)
    float* ws = d.workspace + (size_t)blockIdx.z * d.M * d.N;
#pragma unroll 2
    for (int ps = 0; ps < (WTM * C4 + 63) / 64; ++ps) {
      const int idx = ps * 64 + lane;
      const int row = idx / C4, col = (idx - row * C4) * 4;
      const int m = wm0 + row, n = wn0 + col;
      if (idx < WTM * C4 && m < d.M && n < d.N) *(float4*)(ws + (size_t)m * d.N + n) = *(const float4*)(sC + row * LDW + col);
    }
    return;
  }
  if constexpr (WTN % 32 == 0) {  // GEGLU / QKV epilogues address 32-column blocks (one value|gate block, head-aligned q/k/v)
  if (d.epi == MVD_EPI_GEGLU || d.epi == MVD_EPI_QKV) {
  const int wn0_tile = wn0;
  float* const sC_tile = sC;
  // a wave tile is WTN / 32 such blocks (gemm_kernel: one; gemm_ws_kernel<128, 128, 2, 2>: two), each handled on its own
#pragma unroll 1
  for (int jb = 0; jb < WTN / 32; ++jb) {
  const int wn0 = wn0_tile + jb * 32;
  float* const sC = sC_tile + jb * 32;
  if (wn0 >= d.N) break;
  if (d.epi == MVD_EPI_GEGLU) {   // block = 16 value columns | 16 gate columns
    const int ocol0 = (wn0 >> 5) * 16;
    const int half = d.N >> 1;
    // (rolled chunk loops, column operands loaded once: see MVD_EPI_STORE below)
    const int q = (lane & 3) * 4, col = ocol0 + q;
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f), sg = sv, bv = sv, bg = sv;
    if (lnf) {
      sv = *(const float4*)(d.ln_colsum + col);
      sg = *(const float4*)(d.ln_colsum + half + col);
    }
    if (d.bias) {
      bv = *(const float4*)(d.bias + col);
      bg = *(const float4*)(d.bias + half + col);
    }
    // Two passes like MVD_EPI_STORE below: every chunk's value first (LDS reads + arithmetic, no global access), then all the stores.
    // As one rolled load - compute - store loop the compiler put `s_waitcnt vmcnt(0)` at the loop head (the bias / column-sum loads merge
    // with the loop's stores on the back edge), i.e. every chunk waited for the ACKNOWLEDGEMENT of the previous chunk's stores -- ~1 300
    // cycles when all CUs store at once, four times per 64-row wave tile, about half of this epilogue (round 4, ISA inspection).
    // No run-time branch inside the chunk loop: the LayerNorm fold is a compile-time flag of the lambda and an absent bias adds the zero
    // vector (exact) -- with `if (lnf)` / `if (d.bias)` per chunk every chunk was a chain of small basic blocks, each waiting for its
    // own LDS reads.
    constexpr int NCH = WTM / 16;
    float4 gv[NCH];
    const float scale = d.acc_scale;
    auto geglu_values = [&](auto lnf_c) {
      constexpr bool LNF = decltype(lnf_c)::value;
#pragma unroll
      for (int ps = 0; ps < NCH; ++ps) {
        const int row = ps * 16 + (lane >> 2);
        float4 v = *(const float4*)(sC + row * LDW + q);
        float4 g = *(const float4*)(sC + row * LDW + 16 + q);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
        if (LNF) {
          const float mean = sR[row * 2], rstd = sR[row * 2 + 1];
          v.x = (v.x - mean * sv.x) * rstd; v.y = (v.y - mean * sv.y) * rstd; v.z = (v.z - mean * sv.z) * rstd; v.w = (v.w - mean * sv.w) * rstd;
          g.x = (g.x - mean * sg.x) * rstd; g.y = (g.y - mean * sg.y) * rstd; g.z = (g.z - mean * sg.z) * rstd; g.w = (g.w - mean * sg.w) * rstd;
        }
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        g.x += bg.x; g.y += bg.y; g.z += bg.z; g.w += bg.w;
        gelu_erf4(g.x, g.y, g.z, g.w);        // (packed polynomial: common.hpp)
        v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
        gv[ps] = v;
      }
    };
    if (lnf) geglu_values(std::integral_constant<bool, true>{});
    else geglu_values(std::integral_constant<bool, false>{});
#pragma unroll
    for (int ps = 0; ps < NCH; ++ps) {
      const int m = wm0 + ps * 16 + (lane >> 2);
      if (m < d.M) {
        if (d.out) *(float4*)(d.out + (size_t)m * d.ldo + col) = gv[ps];
        if (d.out_sp) store_sp4((u16*)d.out_sp, (size_t)m, d.ldp, col, gv[ps].x, gv[ps].y, gv[ps].z, gv[ps].w);
      }
    }
    continue;
  }
  {                               // MVD_EPI_QKV: a 32-column aligned block lies inside one of q / k / v
    const int C = d.heads * d.dhead;
    const int which = wn0 / C;
    if (which < 2) {
      const int dq = mvd_attn_dpad(d.dhead);
      u16* ph = (u16*)(which == 0 ? d.q_hi : d.k_hi);
      u16* pl = (u16*)(which == 0 ? d.q_lo : d.k_lo);
      const int col = (lane & 7) * 4, n = wn0 + col;
      const int cc = n - which * C;
      const int head = cc / d.dhead, dd = cc - head * d.dhead;
      const float qs = which == 0 ? d.qscale : 1.0f;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), cs = bb;
      if (d.bias) bb = *(const float4*)(d.bias + n);        // in_proj bias (nn.MultiheadAttention, timm qkv_bias); SD attention has none
      if (lnf) cs = *(const float4*)(d.ln_colsum + n);
      // (values of every chunk first, then all the stores: see the GEGLU epilogue above)
      constexpr int NCQ = WTM / 8;
      float4 qv[NCQ];
      const float scale = d.acc_scale;
      auto qk_values = [&](auto lnf_c) {
        constexpr bool LNF = decltype(lnf_c)::value;
#pragma unroll
        for (int ps = 0; ps < NCQ; ++ps) {
          const int row = ps * 8 + (lane >> 3);
          float4 v = *(const float4*)(sC + row * LDW + col);
          v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
          if (LNF) {
            const float mean = sR[row * 2], rstd = sR[row * 2 + 1];
            v.x = (v.x - mean * cs.x) * rstd; v.y = (v.y - mean * cs.y) * rstd; v.z = (v.z - mean * cs.z) * rstd; v.w = (v.w - mean * cs.w) * rstd;
          }
          qv[ps] = make_float4((v.x + bb.x) * qs, (v.y + bb.y) * qs, (v.z + bb.z) * qs, (v.w + bb.w) * qs);
        }
      };
      if (lnf) qk_values(std::integral_constant<bool, true>{});
      else qk_values(std::integral_constant<bool, false>{});
#pragma unroll
      for (int ps = 0; ps < NCQ; ++ps) {
        const int m = wm0 + ps * 8 + (lane >> 3);
        if (m < d.M) {
          const int b = m / d.L, tok = m - b * d.L;
          const size_t idx = ((size_t)(b * d.heads + head) * d.Lpad + tok) * dq + dd;
          store_planes4(ph, pl, idx, qv[ps].x, qv[ps].y, qv[ps].z, qv[ps].w);
        }
      }
    } else {                      // V^T: each lane takes 4 consecutive tokens of one channel (8-byte stores, keys contiguous)
      const int dv = (d.dhead + 15) & ~15;
      const int col = lane & 31, rsel = lane >> 5;
      const int cc = wn0 + col - 2 * C;
      const int head = cc / d.dhead, dd = cc - head * d.dhead;
      const float bv = d.bias ? d.bias[wn0 + col] : 0.f;
      const float csv = lnf ? d.ln_colsum[wn0 + col] : 0.f;
      constexpr int NCV = WTM / 8;
      float4 vv[NCV];
      const float scale = d.acc_scale;
      auto vt_values = [&](auto lnf_c) {
        constexpr bool LNF = decltype(lnf_c)::value;
#pragma unroll
        for (int ps = 0; ps < NCV; ++ps) {
          const int row = (ps * 2 + rsel) * 4;
          float t4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            t4[i] = sC[(row + i) * LDW + col] * scale;
            if (LNF) t4[i] = (t4[i] - sR[(row + i) * 2] * csv) * sR[(row + i) * 2 + 1];
            t4[i] += bv;
          }
          vv[ps] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      };
      if (lnf) vt_values(std::integral_constant<bool, true>{});
      else vt_values(std::integral_constant<bool, false>{});
#pragma unroll
      for (int ps = 0; ps < NCV; ++ps) {
        const int m = wm0 + (ps * 2 + rsel) * 4;
        if (m < d.M) {
          const int b = m / d.L, tok = m - b * d.L;
          const size_t idx = ((size_t)(b * d.heads + head) * dv + dd) * d.Lpad + tok;
          store_planes4((u16*)d.vt_hi, (u16*)d.vt_lo, idx, vv[ps].x, vv[ps].y, vv[ps].z, vv[ps].w);
        }
      }
    }
  }
  }   // 32-column blocks
  MVD_STAMP_AT(d, wave, 9);
  MVD_STAMP_AT(d, wave, 6);
  MVD_STAMP_AT(d, wave, 7);
  return;
  }
  }
  // MVD_EPI_STORE, in two passes over the wave tile.  A 64-lane chunk is RPC = 64 / C4 whole rows of C4 16-byte columns (80-column wave
  // tiles: 3 rows on 60 lanes), so a lane keeps its column for the whole tile: no division per chunk, the bias / column scale are
  // loaded once, every pointer advances by a constant.
  //   pass 1: epilogue arithmetic on the staged accumulators, final values back into the LDS staging tile; the residual (and the per-view
  //           bias) of chunk ps + 1 is requested before chunk ps is computed; no global store;
  //   pass 2: LDS -> global (fp32 and / or planes); no global load.
  // History (s_memtime stamps, tools/probes/ws_stamp.py; 32x80 wave tile of a 128x80 workgroup tile): one pass, load - compute - store per
  // chunk, everything unrolled and every option (bias, per-view bias, 3 activations, column scale, residual, fp32 / planes outputs) decided
  // at run time per chunk: 14.4 k cycles -- more than the whole k-loop of a K = 320 GEMM -- and ~100 KiB of code per kernel.  Two causes:
  // (1) the stores are conditional, so the compiler cannot count them and waits vmcnt(0) for a load issued after them, i.e. for the
  // acknowledgement of the previous chunk's stores, once per chunk; (2) ~100 VALU / scalar-branch instructions per chunk with ONE
  // wavefront per SIMD to issue them.
  constexpr int RPC = 64 / C4;
  constexpr int NPS = (WTM + RPC - 1) / RPC;
  const int lrow = lane / C4, lcol = (lane - lrow * C4) * 4;
  const int n = wn0 + lcol;
  const int mrow0 = wm0 + lrow;
  const bool lane_ok = lane < RPC * C4 && n + 3 < d.n_store;
  float* const sL = sC + lrow * LDW + lcol;                  // the lane's four values of chunk 0; chunk ps: + ps * RPC * LDW
  const int rows_ok = min(WTM - lrow, d.M - mrow0);          // chunk ps is valid for this lane iff ps * RPC < rows_ok
  auto pass1 = [&](auto act_c, auto res_c, auto bb_c) {
    constexpr int ACT = decltype(act_c)::value;
    constexpr bool HAS_RES = decltype(res_c)::value, HAS_BB = decltype(bb_c)::value;
    // Chunks are processed in GROUPS: all residual / per-view bias requests of a group first, then its arithmetic -- one exposed
    // round trip per group (~850 cycles when all 256 CUs reach their epilogues together; a chunk's arithmetic is ~100).  The requests and
    // the LDS reads are UNCONDITIONAL (lanes / chunks outside the tile read the zero page / the lane's first chunk) so that the compiler can
    // batch and count them: a load under a branch forces vmcnt(0).  No value is carried from one group to the next (a register pipeline
    // across the back edge of the rolled loop makes the compiler rotate registers behind a vmcnt(0)).
    constexpr int GRP = NPS <= 11 ? NPS : (NPS + 1) / 2, NGRP = (NPS + GRP - 1) / GRP;
    const float scale = d.acc_scale;
    const bool has_bias = d.bias != nullptr, has_cs = d.colscale != nullptr;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (has_bias && lane_ok) b = *(const float4*)(d.bias + n);
    if (has_cs && lane_ok) cs = *(const float4*)(d.colscale + n);
    // (the activation variants are compiled for "residual and per-view bias present"; an absent operand reads the zero page)
    const float* const zero = (const float*)g_zero_page;
    const bool use_res = HAS_RES && d.res != nullptr, use_bb = HAS_BB && d.bias_b != nullptr;
    // (one pointer select per lane, outside the loops; inside, only integer offsets are selected -- a select between two POINTERS in
    //  the loop body is compiled into control flow with a load on either side)
    const bool lane_any = lane_ok && rows_ok > 0;
    const float* const rbase = use_res && lane_any ? d.res + (size_t)mrow0 * d.ldr + n : zero;
    const size_t rstep = use_res && lane_any ? (size_t)RPC * d.ldr : 0;
    const float* const bbase = use_bb && lane_any ? d.bias_b + n : zero;
    const int rpb = use_bb ? d.rows_per_batch : 1, ldbb = use_bb && lane_any ? d.ldbb : 0;
#pragma unroll 1
    for (int g = 0; g < NGRP; ++g) {
      float4 qr[GRP], qb[GRP], qv[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int ps = g * GRP + j;
        const bool ok = lane_ok && ps * RPC < rows_ok;
        const int pc = ok ? ps : 0;                  // chunks past the tile re-read the lane's first chunk
        if (HAS_RES) qr[j] = *(const float4*)(rbase + (size_t)pc * rstep);
        if (HAS_BB) qb[j] = *(const float4*)(bbase + (size_t)((mrow0 + pc * RPC) / rpb) * ldbb);
        qv[j] = *(const float4*)(sL + pc * (RPC * LDW));
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int ps = g * GRP + j;
        float4 v = qv[j];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (has_bias) {
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (HAS_BB) {
          v.x += qb[j].x; v.y += qb[j].y; v.z += qb[j].z; v.w += qb[j].w;
        }
        if (ACT == MVD_ACT_GELU) {
          gelu_erf4(v.x, v.y, v.z, v.w);
        } else if (ACT != MVD_ACT_NONE) {
          v.x = apply_act(v.x, ACT); v.y = apply_act(v.y, ACT); v.z = apply_act(v.z, ACT); v.w = apply_act(v.w, ACT);
        }
        if (has_cs) {
          v.x *= cs.x; v.y *= cs.y; v.z *= cs.z; v.w *= cs.w;
        }
        if (HAS_RES) {
          v.x += qr[j].x; v.y += qr[j].y; v.z += qr[j].z; v.w += qr[j].w;
        }
        if (lane_ok && ps * RPC < rows_ok) *(float4*)(sL + ps * (RPC * LDW)) = v;
      }
    }
  };
  {
    using std::integral_constant;
    const integral_constant<bool, true> yes{};
    const integral_constant<bool, false> no{};
    if (d.act == MVD_ACT_NONE) {
      if (d.res) {
        if (d.bias_b) pass1(integral_constant<int, MVD_ACT_NONE>{}, yes, yes);
        else pass1(integral_constant<int, MVD_ACT_NONE>{}, yes, no);
      } else {
        if (d.bias_b) pass1(integral_constant<int, MVD_ACT_NONE>{}, no, yes);
        else pass1(integral_constant<int, MVD_ACT_NONE>{}, no, no);
      }
    } else if (d.act == MVD_ACT_SILU) pass1(integral_constant<int, MVD_ACT_SILU>{}, yes, yes);
    else if (d.act == MVD_ACT_GELU) pass1(integral_constant<int, MVD_ACT_GELU>{}, yes, yes);
    else pass1(integral_constant<int, MVD_ACT_QUICKGELU>{}, yes, yes);
  }
  if (d.n_store & 3 || d.n_store < d.N) {      // ragged n_store edge: element by element (load, compute, store)
    if (lane < RPC * C4 && n < d.N && n + 3 >= d.n_store) {
#pragma unroll 1
      for (int ps = 0; ps * RPC < rows_ok; ++ps)
#pragma unroll 1
        for (int e = 0; e < 4; ++e) epi_store_elem(d, mrow0 + ps * RPC, n + e, sL[ps * (RPC * LDW) + e]);
    }
  }
  MVD_STAMP_AT(d, wave, 9);
  if (lane_ok) {
    float* po = d.out ? d.out + (size_t)mrow0 * d.ldo + n : nullptr;
    u16* psp = d.out_sp ? (u16*)d.out_sp + sp_index((size_t)mrow0, d.ldp, n) : nullptr;
    const size_t ostep = (size_t)RPC * d.ldo, sstep = (size_t)RPC * 2 * d.ldp;
#pragma unroll 1
    for (int p0 = 0; p0 * RPC < rows_ok; p0 += 4) {
      float4 f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = *(const float4*)(sL + ((p0 + j) * RPC < rows_ok ? p0 + j : 0) * (RPC * LDW));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if ((p0 + j) * RPC >= rows_ok) break;
        if (po) {
          *(float4*)po = f[j];
          po += ostep;
        }
        if (psp) {
          uint32_t h0, l0, h1, l1;
          split_op16x2(f[j].x, f[j].y, h0, l0);
          split_op16x2(f[j].z, f[j].w, h1, l1);
          *(uint2*)psp = make_uint2(h0, h1);
          *(uint2*)(psp + 32) = make_uint2(l0, l1);
          psp += sstep;
        }
      }
    }
  }
  MVD_STAMP_AT(d, wave, 6);
#ifdef MVD_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MVD_STAMP_AT(d, wave, 7);
  if (d.rs_out) {
    // per-row {sum, sum of squares} of the stored values over this wave tile's columns -> slot wn0 / WTN of the row (a LayerNorm folded
    // into the consumer GEMM sums the slots in order: deterministic, no atomics).  One lane per row, 16-byte LDS reads.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int ncol = min(WTN, d.n_store - wn0);
    for (int r = lane; r < WTM && wm0 + r < d.M; r += 64) {
      float s1 = 0.f, q1 = 0.f;
      for (int c = 0; c + 3 < ncol; c += 4) {
        const float4 v = *(const float4*)(sC + r * LDW + c);
        s1 += (v.x + v.y) + (v.z + v.w);
        q1 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
      *((float2*)d.rs_out + (size_t)(wm0 + r) * d.rs_ld + wn0 / WTN) = make_float2(s1, q1);
    }
    if (m0 == 0 && n0 == 0 && wave == 0 && lane == 0) d.rs_count[0] = (d.n_store + WTN - 1) / WTN;
  }
  if (d.gn_stats) {
    // GroupNorm statistics of the tensor just produced, for the GroupNorm that consumes it (mvd_groupnorm_from_stats): one lane
    // per column sums its 16-row slabs in row order, the first lane of every (group, slab) fragment adds up its columns in
    // column order and hands the pair to the integer atomics.  (The wave owns its staging tile: LDS ops of one wave are ordered.)
    const int cg = d.n_store / d.gn_groups;
    const int jmax = cg < 64 ? cg : 64;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int c0 = 0; c0 < WTN; c0 += 64) {
      const int col = c0 + lane, n = wn0 + col;
      const bool okc = col < WTN && n < d.n_store;
      const int gidx = okc ? n / cg : 0, pos = okc ? n - gidx * cg : 0;
      const bool leader = okc && (pos == 0 || lane == 0);
      int len = 0;
      if (leader) {
        len = cg - pos;
        if (len > 64 - lane) len = 64 - lane;
        if (len > WTN - col) len = WTN - col;
        if (len > d.n_store - n) len = d.n_store - n;
      }
      // images at least as tall as the wave tile (gn_hw % WTM == 0): one pair of atomics per wave tile and group fragment -- the
      // 16-row slabs are summed in row order first; shorter images: one pair per slab
      const bool whole = d.gn_hw % WTM == 0;
      float s1 = 0.f, q1 = 0.f;
#pragma unroll 1
      for (int sl = 0; sl < WTM / 16; ++sl) {
        const int ms = wm0 + sl * 16;
        if (ms >= d.M) break;
        if (!whole) s1 = q1 = 0.f;
        if (okc) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = sC[(sl * 16 + r) * LDW + col];
            s1 += v;
            q1 += v * v;
          }
        }
        if (whole && sl + 1 < WTM / 16 && ms + 16 < d.M) continue;
        float ss = s1, qq = q1;
        for (int j = 1; j < jmax; ++j) {
          const float ts = __shfl_down(s1, j, 64), tq = __shfl_down(q1, j, 64);
          if (j < len) {
            ss += ts;
            qq += tq;
          }
        }
        if (leader) gn_stats_add(d.gn_stats, ms / d.gn_hw, gidx, d.gn_groups, ss, qq);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ main kernel
template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
  // Counted wait on this wave's own DMA queue plus a full wait on its LDS reads, then the workgroup barrier, as ONE asm
  // statement with a memory clobber.  vmcnt(N): the compiler does not drain the DMA queue to 0 (as __syncthreads would
  // with LDS-DMA in flight).  lgkmcnt(0): the fragment reads issued before the barrier must have RETURNED before any
  // other wave is released to overwrite the buffer (next DMA, or the epilogue staging tile) -- the compiler is free to
  // sink the MFMAs that consume them, and with them its own lgkmcnt wait, below the barrier.
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory");
}

// Instruction-mix hint for one pipelined k-tile: SLOTS groups of [a few MFMAs, one memory instruction]; the first LPS
// memory slots are the LDS-DMA issues (longest latency), the rest the LDS fragment reads of the next k-tile.
template <int G, int SLOTS, int NM, int LPS>
__device__ __forceinline__ void sched_pattern() {
  if constexpr (G < SLOTS) {
    constexpr int mf = NM * (G + 1) / SLOTS - NM * G / SLOTS;
    if constexpr (mf > 0) __builtin_amdgcn_sched_group_barrier(0x008, mf, 0);   // MFMA
    if constexpr (G < LPS)
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        // VMEM read (the LDS-DMA)
    else
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        // DS read
    sched_pattern<G + 1, SLOTS, NM, LPS>();
  }
}

template <int J, int N, class F>
__device__ __forceinline__ void unroll_steps(F&& f) {
  if constexpr (J < N) {
    f(std::integral_constant<int, J>{});
    unroll_steps<J + 1, N>(f);
  }
}

// Consumer wavefronts of the role-split kernels: the NR fragment reads of the NEXT k-tile go out behind the first MFMAs of this one (one
// read per MFMA), so that every one of them has returned long before the `lgkmcnt(0)` + barrier that ends the iteration -- spread evenly
// over the k-tile (sched_pattern) the last read is a few MFMAs old when the wave reaches that wait, and the matrix pipe drains behind it.
template <int G, int NR, int NM>
__device__ __forceinline__ void sched_reads_early() {
  if constexpr (G < NR) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    sched_reads_early<G + 1, NR, NM>();
  } else if constexpr (NM > NR) {
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
  }
}

// (second launch bound: the register-staged 8-wave tile must leave room for two workgroups per CU = 4 wavefronts per SIMD)
template <int BM, int BN, int WM, int WN, int NS, int AMODE, int STAGES>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8 && STAGES == 8) ? 4 : 1) void gemm_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int A_GRAN = BM / 8, B_GRAN = BN / 8;     // 1 KiB granules (A: 8 rows x 128 B; packed B: one hi or lo fragment image)
  constexpr int AI = A_GRAN / NW, BI = (B_GRAN + NW - 1) / NW;   // granules per wave per k-tile
  constexpr int B_GRAN_P = BI * NW;                   // B granules rounded up to a multiple of the wave count: every wave issues
                                                      // the same number of DMAs (counted vmcnt); the extra ones copy the zero page
  constexpr int STAGE = (A_GRAN + B_GRAN_P) * 1024;
  constexpr int LPS = AI + BI;                        // DMA instructions per wave per stage
  constexpr int LDW = WTN + 4;                        // fp32 pitch of the epilogue staging tile
  constexpr int C4 = WTN / 4;                         // float4 columns of a wave tile row
  constexpr int EPI_BYTES = NW * WTM * LDW * 4;
  constexpr bool RING = STAGES == 6 || STAGES == 7;    // 6 / 7 = register-pipelined loop over a DEEP ring of LDS buffers (<= 4 / <= 8)
  constexpr bool PIPE = STAGES == 3 || RING;           // 3 = register-pipelined loop (two LDS buffers)
  constexpr bool STAG = STAGES == 4 || STAGES == 5;    // 4 / 5 = staggered wave groups with three / four LDS buffers
  constexpr bool RSTG = STAGES == 8;                    // 8 = REGISTER-staged delivery: global_load_dwordx4 -> VGPR -> ds_write_b128, two LDS buffers
  constexpr int TAB_BYTES = AMODE != MVD_A_DENSE ? BM * 9 * 4 : 0;
  constexpr int LNR_BYTES = AMODE == MVD_A_DENSE ? BM * 8 : 0;      // {mean, rstd} of the tile's rows (LayerNorm fold: dense problems)
  constexpr int RING_FIT = (160 * 1024 - TAB_BYTES - LNR_BYTES) / STAGE;   // a workgroup may own the whole 160 KiB of its CU
  constexpr int RING_WANT = STAGES == 6 ? 4 : 8;
  constexpr int NBUF = STAG ? STAGES - 1 : (RING ? (RING_WANT < RING_FIT ? RING_WANT : RING_FIT) : 2);
  constexpr int LEAD = NBUF - 1;                       // staggered loop: k-tiles staged ahead of the one being read
  constexpr int SMEM = NBUF * STAGE > EPI_BYTES ? NBUF * STAGE : EPI_BYTES;
  static_assert(NBUF >= 2 && SMEM + TAB_BYTES + LNR_BYTES <= 160 * 1024, "LDS budget");
  static_assert(!STAG || NW == 8, "the staggered loop pairs the two wavefronts of each SIMD: 8-wave tiles only");
  static_assert(A_GRAN % NW == 0, "A granules must divide evenly over the waves");
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && B_GRAN % 2 == 0, "wave tiles are made of 16x16 MFMA tiles");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM + TAB_BYTES + LNR_BYTES];

  const mvd_gemm_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  MVD_STAMP_AT(d, wave, 0);
  // XCD-aware tile mapping: workgroup b runs on XCD b % 8 (each XCD has a private 4 MiB L2).  Give every XCD a
  // contiguous range of output tiles in n-fastest order, so the n-tiles that re-read one A row panel (and the
  // neighbouring m-tiles that share the conv halo) hit the same L2 instead of 8 different ones.
  int tile;
  {
    const int nb = p.tiles_n * p.tiles_m, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // n-fastest: an XCD re-uses one A row panel across its n-tiles (and streams all of W);
  // m-fastest: an XCD keeps a W column panel resident and streams A -- chosen per problem by bytes moved.
  const int m0 = (p.m_fastest ? tile % p.tiles_m : tile / p.tiles_n) * BM;
  const int n0 = (p.m_fastest ? tile / p.tiles_m : tile % p.tiles_n) * BN;
  const int kt0 = blockIdx.z * p.kt_per_split;
  const int kt1 = min(p.nk, kt0 + p.kt_per_split);
  const int nkt = kt1 - kt0;
  // LayerNorm fold (mvd_gemm_desc.ln_stats): mean / rstd of the tile's rows from the producer's slots, one thread per row, called right
  // after the prologue's DMAs are in flight (the loads' round trips hide behind the first k-tile's) and read by the epilogue -- the
  // k-loop's barriers order the two.
  float* s_rows = (float*)(smem + SMEM + TAB_BYTES);
  auto ln_gather_rows = [&]() {
    if (AMODE == MVD_A_DENSE && d.ln_stats != nullptr && tid < BM) {
      const float2 st = m0 + tid < d.M ? ln_row_stats(d, m0 + tid) : make_float2(0.f, 0.f);
      s_rows[tid * 2] = st.x;
      s_rows[tid * 2 + 1] = st.y;
    }
  };

  // ---- per-lane staging roles.  Lane l of a granule fills slot l: row r = l>>3 (of 8), stored chunk l&7 holds source
  //      chunk cc = (l&7) ^ f(R), f(R) = (R>>1) & 7 with R the row inside its 16-row MFMA block.
  const int gr = lane >> 3;
  const u16* zero = (const u16*)g_zero_page;

  const u16* a_src[AI];     // dense: per A granule source row base (k = 0, + this lane's chunk)
  bool a_ok[AI];
  int a_tab[AI], a_chunk[AI];   // conv: LDS index of this lane's row in the tap table, chunk offset inside the 128-byte line
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int gi = wave + i * NW;            // A granule index = 8-row group of the block tile
    const int R = (gi & 1) * 8 + gr;
    const int gc = (lane & 7) ^ ((R >> 1) & 7);
    const int m = m0 + gi * 8 + gr;
    a_ok[i] = m < d.M;
    a_src[i] = (const u16*)d.A + (size_t)(a_ok[i] ? m : 0) * 2 * d.lda + gc * 8;
    a_tab[i] = (gi * 8 + gr) * 9;
    a_chunk[i] = gc * 8;
  }
  // conv: source offset (u16 units from d.A, channel 0) of every (tile row, filter tap), -1 where the tap falls into
  // the zero padding or the row is outside M.  Filled once per workgroup; the k loop reads one entry per granule.
  int* s_tab = (int*)(smem + SMEM);
  if (AMODE != MVD_A_DENSE) {
    const int hw = d.Hout * d.Wout;
    for (int e = tid; e < BM * 9; e += NW * 64) {
      const int row = e / 9, tap = e - row * 9;
      const int m = m0 + row;
      int off = -1;
      if (m < d.M) {
        const int b = m / hw;
        const int rem = m - b * hw;
        const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
        const int ky = tap / 3, kx = tap - ky * 3;
        int iy, ix;
        bool ok;
        if (d.upsample) {
          const int uy = oy + ky - 1, ux = ox + kx - 1;
          ok = uy >= 0 && uy < d.Hout && ux >= 0 && ux < d.Wout;
          iy = uy >> 1;
          ix = ux >> 1;
        } else {
          iy = oy * d.stride + ky - (d.no_pad_tl ? 0 : 1);
          ix = ox * d.stride + kx - (d.no_pad_tl ? 0 : 1);
          ok = iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win;
        }
        if (ok) off = ((b * d.Hin + iy) * d.Win + ix) * 2 * d.Cin;
      }
      s_tab[e] = off;
    }
    __syncthreads();
  }
  const u16* b_src[BI];     // per B granule: its source at kt = 0 (+ this lane's 16 bytes) or null (-> zero page)
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int gi = wave + i * NW;            // B granule index: packed weight = image gi & 1 of micro-tile gi >> 1; planes = 8-row group
    const int nt = (n0 >> 4) + (gi >> 1);
    if (d.b_mode == MVD_B_PLANES) {           // B rows are rows of an activation matrix in split planes (same 128-byte lines as A)
      const int R = (gi & 1) * 8 + gr;
      const int gc = (lane & 7) ^ ((R >> 1) & 7);
      const int n = n0 + gi * 8 + gr;
      b_src[i] = (gi < B_GRAN && n < d.N) ? (const u16*)d.Wp + (size_t)n * 2 * d.ldb + gc * 8 : nullptr;
    } else {
      b_src[i] = (gi < B_GRAN && nt < p.nt16) ? (const u16*)d.Wp + (size_t)nt * 1024 + (gi & 1) * 512 + lane * 8 : nullptr;
    }
  }
  // elements between consecutive k-tiles: packed weight = one row of micro-tiles; planes = the next 128-byte line of the row
  const size_t b_kstride = d.b_mode == MVD_B_PLANES ? (size_t)64 : (size_t)p.nt16 * 1024;

  // Running DMA sources: every stage() call moves one k-tile forward.  Dense A and the packed weights advance a
  // pointer (rows / weight tiles outside the problem sit on the zero page with step 0).  Conv K order is
  // (32-channel block, tap, channel): the 9 taps of one channel block are consecutive k-tiles, so the 3x3 neighbourhood
  // re-reads of a 128-byte pixel line happen back to back and hit L2 (tap-major order re-fetched the whole image 9 times
  // from the memory side: 9x the algorithmic A bytes in FETCH_SIZE).
  const u16* a_cur[AI];
  int a_step[AI];
  int a_off[AI];                              // conv: table entry of the tap staged next
  int c_tap = 0, c_cb = 0;                    // conv: tap and channel block of the k-tile staged next (uniform)
  if (AMODE == MVD_A_DENSE) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      a_cur[i] = a_ok[i] ? a_src[i] + (size_t)kt0 * 64 : zero;
      a_step[i] = a_ok[i] ? 64 : 0;
    }
  } else {
    c_cb = kt0 / 9;
    c_tap = kt0 - c_cb * 9;
#pragma unroll
    for (int i = 0; i < AI; ++i) a_off[i] = s_tab[a_tab[i] + c_tap];
  }
  const u16* b_cur[BI];
  size_t b_step[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    b_cur[i] = b_src[i] ? b_src[i] + (size_t)kt0 * b_kstride : zero;
    b_step[i] = b_src[i] ? b_kstride : 0;
  }

  auto stage = [&](int buf) {                 // DMA the next k-tile (consecutive calls walk kt0, kt0+1, ...)
    unsigned char* sbase = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const u16* src;
      if (AMODE == MVD_A_DENSE) {
        src = a_cur[i];
        a_cur[i] += a_step[i];
      } else {
        src = a_off[i] >= 0 ? (const u16*)d.A + (unsigned)(a_off[i] + c_cb * 64 + a_chunk[i]) : zero;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sbase + (wave + i * NW) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)b_cur[i],
          (__attribute__((address_space(3))) void*)(sbase + (A_GRAN + wave + i * NW) * 1024), 16, 0, 0);
      b_cur[i] += b_step[i];
    }
  };
  auto advance_tap = [&]() {                  // conv bookkeeping after each stage(): next tap, prefetch its table entries
    if (AMODE != MVD_A_DENSE) {
      if (++c_tap == 9) {
        c_tap = 0;
        ++c_cb;
      }
#pragma unroll
      for (int i = 0; i < AI; ++i) a_off[i] = s_tab[a_tab[i] + c_tap];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = lane&15 of a 16-row block (2 granules), hi chunk = lane>>4, lo chunk = 4 + (lane>>4)
  const int frow = lane & 15;
  const int fsw = (frow >> 1) & 7;
  const int fbase = (frow >> 3) * 1024 + (frow & 7) * 128;
  const int foff_hi = fbase + (((lane >> 4)) ^ fsw) * 16;
  const int foff_lo = fbase + ((4 + (lane >> 4)) ^ fsw) * 16;

  auto mfma_tile = [&](const op16x8 (&ah)[TM], const op16x8 (&al)[TM], const op16x8 (&bh)[TN], const op16x8 (&bl)[TN]) {
    // term-major order: consecutive MFMAs hit different accumulators (no back-to-back dependency); every accumulator
    // still receives lo*lo, lo*hi, hi*lo, hi*hi in that order per k-tile (the summation order is part of the numerics).
    if (NS == 4) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(al[i], bl[j], acc[i][j], 0, 0, 0);
    }
    if (NS >= 3) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(ah[i], bl[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(ah[i], bh[j], acc[i][j], 0, 0, 0);
  };
  // B fragments in LDS: a packed micro-tile is already two fragment images (lane l at byte 16 l); planes are laid out like A
  const int boff_hi = d.b_mode == MVD_B_PLANES ? foff_hi : lane * 16;
  const int boff_lo = d.b_mode == MVD_B_PLANES ? foff_lo : 1024 + lane * 16;
  auto read_frags = [&](int buf, op16x8 (&ah)[TM], op16x8 (&al)[TM], op16x8 (&bh)[TN], op16x8 (&bl)[TN]) {
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + A_GRAN * 1024;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *(const op16x8*)(sA + (wm * TM + i) * 2048 + foff_hi);
      if (NS >= 3) al[i] = *(const op16x8*)(sA + (wm * TM + i) * 2048 + foff_lo);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *(const op16x8*)(sB + (wn * TN + j) * 2048 + boff_hi);
      if (NS >= 3) bl[j] = *(const op16x8*)(sB + (wn * TN + j) * 2048 + boff_lo);
    }
  };

  if (STAG) {
    // ---- staggered loop, three LDS buffers.  Phases ph = 0 .. 2 nkt, one raw barrier each.  Group g (0: waves 0-3, 1: waves
    //      4-7 -- a workgroup's waves are dealt to the 4 SIMDs round-robin, so each group has one wave per SIMD) runs
    //      MEM(t) in phase 2t + g and MFMA(t) in phase 2t + g + 1:
    //        MEM(t)  : issue this wave's DMA share of k-tile t+LEAD into buffer (t+LEAD)%NBUF (its last readers finished two
    //                  phases ago), read the fragments of k-tile t, then wait until this wave's share of k-tile t+1 has landed
    //                  (counted vmcnt: the newer stages stay in flight) -- its first reader is two barriers away.  An LDS-DMA
    //                  round trip is ~1500 cycles even from L2 (tools/probes/dma_probe.hip), longer than one k-tile of MFMAs,
    //                  so LEAD >= 2 stages must be in flight per workgroup
    //        MFMA(t) : the TM x TN x NS MFMAs on the fragments read in the previous phase
    //      so at any time one wave of a SIMD feeds the MFMA pipe while the other one issues memory instructions.
    const int grp = wave >> 2;
    op16x8 ah[TM], al[TM], bh[TN], bl[TN];
    // prologue: k-tiles 0 .. LEAD-1 in flight, k-tile 0 landed for everybody
#pragma unroll
    for (int q = 0; q < LEAD; ++q) {
      if (q < nkt) {
        stage(q);
        advance_tap();
      }
    }
    ln_gather_rows();
    MVD_STAMP_AT(d, wave, 1);
    if (nkt >= LEAD) wait_vm_and_barrier<(LEAD - 1) * LPS>();
    else wait_vm_and_barrier<0>();
    MVD_STAMP_AT(d, wave, 2);
    for (int ph = 0; ph <= 2 * nkt; ++ph) {
      const int u = ph - grp;
      if (u >= 0 && u < 2 * nkt) {
        const int t = u >> 1;
        if ((u & 1) == 0) {
          // MEM(t): stage k-tile t + LEAD, read the fragments of k-tile t, then make sure this wave's share of k-tile t + 1
          // has landed: only the newest LEAD - 1 stages (k-tiles t + 2 .. t + LEAD) may still be in flight
          const bool more = t + LEAD < nkt;
          if (more) {
            stage((t + LEAD) % NBUF);
            advance_tap();
          }
          read_frags(t % NBUF, ah, al, bh, bl);
          if (more) wait_vm_and_barrier<(LEAD - 1) * LPS>();
          else wait_vm_and_barrier<0>();      // tail: drain (at most LEAD - 1 short iterations)
          continue;
        }
        mfma_tile(ah, al, bh, bl);
      }
      asm volatile("s_barrier" ::: "memory");
    }
    __syncthreads();   // the epilogue reuses the stage buffers
  } else if (PIPE) {
    // ---- register-pipelined loop over a ring of NBUF LDS buffers (NBUF = 2: STAGES 3; up to 4 / 8: the RING variants).  While the
    //      MFMAs of k-tile t run out of one fragment register set, the wave reads k-tile t+1 from LDS into the other set and issues
    //      the DMA of k-tile t+NBUF into the buffer that tile t occupied (its fragments are already in registers).  One barrier per
    //      k-tile; NBUF - 1 k-tiles of operands are in flight per workgroup, so a small grid (one workgroup per CU, as the low-resolution
    //      levels of the UNet give) is not bound by one DMA round trip per k-tile: Little's law with 16 KiB in flight per CU and
    //      ~1.5 us from a cold weight to LDS is ~10 GB/s per CU; a ring of 8 lifts that bound 7x.
    op16x8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
#pragma unroll
    for (int q = 0; q < NBUF; ++q) {
      if (q < nkt) {
        stage(q);
        advance_tap();
      }
    }
    ln_gather_rows();
    MVD_STAMP_AT(d, wave, 1);
    if (nkt >= NBUF) wait_vm_and_barrier<(NBUF - 1) * LPS>();   // k-tile 0 landed, the newer ones stay in flight
    else wait_vm_and_barrier<0>();
    MVD_STAMP_AT(d, wave, 2);
    read_frags(0, fah[0], fal[0], fbh[0], fbl[0]);
    int bs = 0, br = NBUF > 1 ? 1 : 0;              // buffer staged next (= it % NBUF), buffer read next (= (it + 1) % NBUF)
    auto step = [&](auto parity, auto steady, int it) {
      constexpr int P = decltype(parity)::value;
      constexpr bool FULL = decltype(steady)::value;   // steady state: no conditions -> one basic block to schedule
      // k-tile it+1 has landed for every wave (the NBUF - 2 newer stages may still fly), and every wave's fragment reads of the
      // buffer of k-tile it have returned
      if (FULL) wait_vm_and_barrier<(NBUF - 2) * LPS>();
      else wait_vm_and_barrier<0>();
      if (FULL || it + NBUF < nkt) stage(bs);
      if (FULL || it + 1 < nkt) read_frags(br, fah[P ^ 1], fal[P ^ 1], fbh[P ^ 1], fbl[P ^ 1]);
      mfma_tile(fah[P], fal[P], fbh[P], fbl[P]);
      if (FULL) {
        constexpr int NM = TM * TN * NS, NR = (TM + TN) * (NS >= 3 ? 2 : 1);
        sched_pattern<0, LPS + NR, NM, LPS>();     // (the conv table reads of advance_tap() follow the pattern)
      }
      if (FULL || it + NBUF < nkt) advance_tap();
      bs = bs + 1 == NBUF ? 0 : bs + 1;
      br = br + 1 == NBUF ? 0 : br + 1;
    };
    using std::integral_constant;
    int it = 0;
    for (; it + NBUF + 1 < nkt; it += 2) {
      step(integral_constant<int, 0>{}, integral_constant<bool, true>{}, it);
      step(integral_constant<int, 1>{}, integral_constant<bool, true>{}, it + 1);
    }
    // at most NBUF + 1 k-tiles remain (`it` is even).  Straight-line on purpose: as a loop with a run-time parity switch
    // the compiler carried the accumulators through AGPR copies on the back edge, and one of them (v_accvgpr_mov of the
    // register the last MFMA had just written) read a stale value in the 64x64 conv instantiation -- every
    // configuration is now cross-checked in tests/test_gpu_ops.py::test_gemm_configurations_agree.
    unroll_steps<0, NBUF + 1>([&](auto j) {
      constexpr int J = decltype(j)::value;
      if (it + J < nkt) step(integral_constant<int, J & 1>{}, integral_constant<bool, false>{}, it + J);
    });
    __syncthreads();   // the epilogue reuses the stage buffers
  } else if (RSTG) {
    // ---- register-staged loop (round 4).  The LDS-DMA instruction that the other loops issue per 1 KiB granule costs 60 - 185 issue
    //      cycles on the wave that issues it; a 64x64 tile gives a wave only 16 MFMAs (256 cycles) per k-tile against 4 such granules, so
    //      the small-tile kernels are DMA-ISSUE bound (12 - 13 % of the matrix pipe, profiles/r03_pmc_mfma.json).  Here every wave loads
    //      its granules with ordinary 16-byte global loads two k-tiles ahead (two register stages of LPS x 4 VGPRs), and writes a k-tile
    //      into the other LDS buffer with ds_write_b128 (same lane-linear 1 KiB image the DMA would have produced) while the current one
    //      is multiplied: ~20 issue cycles per granule instead of ~100.  Small LDS footprint (2 buffers), so several workgroups share
    //      a CU and hide each other's barriers.  Same k / MFMA order as every other loop => bit-identical results.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 regs[2][LPS];
    auto fetch = [&](auto rs_c) {                // issue the loads of the next k-tile into register stage RS (consecutive calls walk kt0, kt0 + 1, ...)
      constexpr int RS = decltype(rs_c)::value;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const u16* src;
        if (AMODE == MVD_A_DENSE) {
          src = a_cur[i];
          a_cur[i] += a_step[i];
        } else {
          src = a_off[i] >= 0 ? (const u16*)d.A + (unsigned)(a_off[i] + c_cb * 64 + a_chunk[i]) : zero;
        }
        regs[RS][i] = *(const u32x4*)src;
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        regs[RS][AI + i] = *(const u32x4*)b_cur[i];
        b_cur[i] += b_step[i];
      }
      advance_tap();
    };
    auto put = [&](auto rs_c, int buf) {
      constexpr int RS = decltype(rs_c)::value;
      unsigned char* sbase = smem + buf * STAGE + lane * 16;
#pragma unroll
      for (int i = 0; i < AI; ++i) *(u32x4*)(sbase + (wave + i * NW) * 1024) = regs[RS][i];
#pragma unroll
      for (int i = 0; i < BI; ++i) *(u32x4*)(sbase + (A_GRAN + wave + i * NW) * 1024) = regs[RS][AI + i];
    };
    using std::integral_constant;
    fetch(integral_constant<int, 0>{});                              // k-tile 0 -> stage 0
    if (nkt > 1) fetch(integral_constant<int, 1>{});                 // k-tile 1 -> stage 1
    ln_gather_rows();
    MVD_STAMP_AT(d, wave, 1);
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    put(integral_constant<int, 0>{}, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    MVD_STAMP_AT(d, wave, 2);
    // iteration it (k-tile it sits in LDS buffer it & 1; register stage j & 1 holds k-tile j): read the fragments, refill the stage k-tile it
    // left with k-tile it + 2, multiply, then move k-tile it + 1 (loaded a whole iteration ago) into the other buffer -- its last readers
    // passed the previous barrier
    auto body = [&](auto par_c, int it) {
      constexpr int P = decltype(par_c)::value;
      op16x8 ah[TM], al[TM], bh[TN], bl[TN];
      read_frags(P, ah, al, bh, bl);
      if (it + 2 < nkt) fetch(integral_constant<int, P>{});
      mfma_tile(ah, al, bh, bl);
      if (it + 1 < nkt) {
        if (it + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS) : "memory");      // k-tile it + 1 landed; it + 2 stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        put(integral_constant<int, P ^ 1>{}, P ^ 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int it = 0; it < nkt; it += 2) {
      body(integral_constant<int, 0>{}, it);
      if (it + 1 < nkt) body(integral_constant<int, 1>{}, it + 1);
    }
  } else {
    // ---- plain two-buffer loop: DMA of k-tile t+1 in flight while tile t is read and multiplied
    stage(0);
    advance_tap();
    ln_gather_rows();
    MVD_STAMP_AT(d, wave, 1);
    wait_vm_and_barrier<0>();
    MVD_STAMP_AT(d, wave, 2);
    int buf = 0;
    for (int it = 0; it < nkt; ++it) {
      if (it + 1 < nkt) {
        stage(buf ^ 1);
        advance_tap();
      }
      op16x8 ah[TM], al[TM], bh[TN], bl[TN];
      read_frags(buf, ah, al, bh, bl);
      mfma_tile(ah, al, bh, bl);
      wait_vm_and_barrier<0>();   // k-tile it+1 landed (all waves); nobody still reads buffer `buf`
      buf ^= 1;
    }
  }

  // ---- epilogue (the final barrier above guarantees nobody still reads the stage buffers; each wave owns a private region)
  MVD_STAMP_AT(d, wave, 3);
  tile_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, lane, wave, AMODE == MVD_A_DENSE && d.ln_stats != nullptr ? s_rows : nullptr);
  MVD_STAMP_AT(d, wave, 8);
}

// ------------------------------------------------------------------------------------------------ wave-specialised GEMM
// gemm_ws_kernel: the same operands, tiles, k order and MFMA order as gemm_kernel (=> bit-identical results), but the eight wavefronts of
// the workgroup have two ROLES: waves 0-3 are CONSUMERS (a CM x CN arrangement, one per SIMD: LDS fragment reads + MFMAs, fragments
// double-buffered in registers) and waves 4-7 are LOADERS (one per SIMD: they issue every LDS-DMA of the ring of NBUF stages and wait for
// them with counted vmcnt).  In gemm_kernel every wavefront issues its share of the DMAs (60-185 cycles of issue each), its fragment reads
// and its MFMAs in ONE in-order instruction stream, and all loop variants saturate at 43-55 % of the matrix pipe; here a SIMD's MFMA stream
// never contains a memory instruction other than its own ds_reads, and the DMA issue of the loader runs beside it (separate issue ports).
// One workgroup barrier per k-tile orders the two roles:
//   before barrier B_t : loaders have waited until THEIR share of k-tile t+1 landed; consumers until their reads of k-tile t returned
//   after  barrier B_t : loaders stage k-tile t+NBUF into the buffer of k-tile t (its fragments sit in registers), then wait for k-tile
//                        t+2 (the NBUF-2 newer stages stay in flight); consumers read the fragments of k-tile t+1 and run the MFMAs of t.
// LM (loader mode): 0 = LDS-DMA (global_load_lds_dwordx4, counted vmcnt); 1 = through REGISTERS: the loader wavefronts issue ordinary
// global_load_dwordx4 for k-tile t + NBUF, keep NBUF - 2 k-tiles of their share in VGPRs (the consumers' register allocation is
// kernel-wide: the loaders have ~160 idle registers) and write a k-tile into its LDS slot with ds_write_b128 one iteration before the
// consumers read it.  Same LDS image, same barriers, same MFMA order => bit-identical; the tuner decides per shape which delivery
// path is faster (LDS-DMA: ~1 KiB per 60-185 issue cycles and wave; ds_write_b128: ~13 cycles per KiB-instruction).
template <int BM, int BN, int CM, int CN, int NS, int AMODE, int LM = 0>
__global__ __launch_bounds__(512) void gemm_ws_kernel(GemmParams p) {
  constexpr int NC = CM * CN, NL = 4;
  static_assert(NC == 4, "four consumer wavefronts (one per SIMD) + four loader wavefronts");
  constexpr int WTM = BM / CM, WTN = BN / CN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int A_GRAN = BM / 8, B_GRAN = BN / 8;
  constexpr int AI = A_GRAN / NL, BI = (B_GRAN + NL - 1) / NL;
  constexpr int B_GRAN_P = BI * NL;
  constexpr int STAGE = (A_GRAN + B_GRAN_P) * 1024;
  constexpr int LPS = AI + BI;
  constexpr int LDW = WTN + 4;
  constexpr int EPI_BYTES = NC * WTM * LDW * 4;
  constexpr int TAB_BYTES = AMODE != MVD_A_DENSE ? BM * 9 * 4 : 0;
  constexpr int LNR_BYTES = AMODE == MVD_A_DENSE ? BM * 8 : 0;
  constexpr int FIT = (160 * 1024 - TAB_BYTES - LNR_BYTES) / STAGE;
  constexpr int NBUF = FIT < 8 ? FIT : 8;      // as deep as the CU's LDS allows: the loader's lead is NBUF - 2 k-tiles (one DMA round trip ~ 2 k-tiles of MFMAs)
  constexpr int SMEM = NBUF * STAGE > EPI_BYTES ? NBUF * STAGE : EPI_BYTES;
  static_assert(NBUF >= 3 && SMEM + TAB_BYTES + LNR_BYTES <= 160 * 1024, "LDS budget");
  static_assert(A_GRAN % NL == 0 && WTM % 16 == 0 && WTN % 16 == 0 && B_GRAN % 2 == 0, "tile geometry");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM + TAB_BYTES + LNR_BYTES];

  const mvd_gemm_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  MVD_STAMP_AT(d, wave, 0);
  int tile;
  {
    const int nb = p.tiles_n * p.tiles_m, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (p.m_fastest ? tile % p.tiles_m : tile / p.tiles_n) * BM;
  const int n0 = (p.m_fastest ? tile / p.tiles_m : tile % p.tiles_n) * BN;
  const int kt0 = blockIdx.z * p.kt_per_split;
  const int nkt = min(p.nk, kt0 + p.kt_per_split) - kt0;
  float* s_rows = (float*)(smem + SMEM + TAB_BYTES);
  int* s_tab = (int*)(smem + SMEM);
  if (AMODE != MVD_A_DENSE) {      // conv: source offset of every (tile row, filter tap), -1 in the zero padding / past M (as gemm_kernel)
    const int hw = d.Hout * d.Wout;
    for (int e = tid; e < BM * 9; e += 512) {
      const int row = e / 9, tap = e - row * 9;
      const int m = m0 + row;
      int off = -1;
      if (m < d.M) {
        const int b = m / hw;
        const int rem = m - b * hw;
        const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
        const int ky = tap / 3, kx = tap - ky * 3;
        int iy, ix;
        bool ok;
        if (d.upsample) {
          const int uy = oy + ky - 1, ux = ox + kx - 1;
          ok = uy >= 0 && uy < d.Hout && ux >= 0 && ux < d.Wout;
          iy = uy >> 1;
          ix = ux >> 1;
        } else {
          iy = oy * d.stride + ky - (d.no_pad_tl ? 0 : 1);
          ix = ox * d.stride + kx - (d.no_pad_tl ? 0 : 1);
          ok = iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win;
        }
        if (ok) off = ((b * d.Hin + iy) * d.Win + ix) * 2 * d.Cin;
      }
      s_tab[e] = off;
    }
    __syncthreads();
  }

  if (wave >= NC) {
    // ================================================================ LOADER wavefronts
    const int lw = wave - NC;
    const int gr = lane >> 3;
    const u16* zero = (const u16*)g_zero_page;
    const u16* a_cur[AI];
    int a_step[AI], a_tab[AI], a_chunk[AI], a_off[AI];
    int c_tap = 0, c_cb = 0;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int gi = lw + i * NL;
      const int R = (gi & 1) * 8 + gr;
      const int gc = (lane & 7) ^ ((R >> 1) & 7);
      const int m = m0 + gi * 8 + gr;
      const bool ok = m < d.M;
      a_cur[i] = ok ? (const u16*)d.A + (size_t)m * 2 * d.lda + gc * 8 + (size_t)kt0 * 64 : zero;
      a_step[i] = ok ? 64 : 0;
      a_tab[i] = (gi * 8 + gr) * 9;
      a_chunk[i] = gc * 8;
    }
    if (AMODE != MVD_A_DENSE) {
      c_cb = kt0 / 9;
      c_tap = kt0 - c_cb * 9;
#pragma unroll
      for (int i = 0; i < AI; ++i) a_off[i] = s_tab[a_tab[i] + c_tap];
    }
    const u16* b_cur[BI];
    size_t b_step[BI];
    {
      const size_t b_kstride = d.b_mode == MVD_B_PLANES ? (size_t)64 : (size_t)p.nt16 * 1024;
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int gi = lw + i * NL;
        const int nt = (n0 >> 4) + (gi >> 1);
        const u16* src;
        if (d.b_mode == MVD_B_PLANES) {
          const int R = (gi & 1) * 8 + gr;
          const int gc = (lane & 7) ^ ((R >> 1) & 7);
          const int n = n0 + gi * 8 + gr;
          src = (gi < B_GRAN && n < d.N) ? (const u16*)d.Wp + (size_t)n * 2 * d.ldb + gc * 8 : nullptr;
        } else {
          src = (gi < B_GRAN && nt < p.nt16) ? (const u16*)d.Wp + (size_t)nt * 1024 + (gi & 1) * 512 + lane * 8 : nullptr;
        }
        b_cur[i] = src ? src + (size_t)kt0 * b_kstride : zero;
        b_step[i] = src ? b_kstride : 0;
      }
    }
    auto stage = [&](int buf) {
      unsigned char* sbase = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const u16* src;
        if (AMODE == MVD_A_DENSE) {
          src = a_cur[i];
          a_cur[i] += a_step[i];
        } else {
          src = a_off[i] >= 0 ? (const u16*)d.A + (unsigned)(a_off[i] + c_cb * 64 + a_chunk[i]) : zero;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sbase + (lw + i * NL) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_cur[i],
                                         (__attribute__((address_space(3))) void*)(sbase + (A_GRAN + lw + i * NL) * 1024), 16, 0, 0);
        b_cur[i] += b_step[i];
      }
      if (AMODE != MVD_A_DENSE) {
        if (++c_tap == 9) {
          c_tap = 0;
          ++c_cb;
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) a_off[i] = s_tab[a_tab[i] + c_tap];
      }
    };
    if constexpr (LM == 1) {
      // ---- register-staged delivery.  RD = NBUF - 2 register stages: stage j % RD holds k-tile j from its issue (iteration j - NBUF,
      //      right after barrier B_{j-NBUF}: slot j % NBUF is free then) until it is written to LDS in iteration j - 2 (before barrier
      //      B_{j-1}, after which the consumers read it).  The loop is unrolled by RD so that every register stage is a compile-time index.
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      constexpr int RD = NBUF - 2;
      u32x4 regs[RD][LPS];
      auto fetch = [&](auto rs_c) {                  // issue the loads of the next k-tile (consecutive calls walk kt0, kt0 + 1, ...)
        constexpr int RS = decltype(rs_c)::value;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const u16* src;
          if (AMODE == MVD_A_DENSE) {
            src = a_cur[i];
            a_cur[i] += a_step[i];
          } else {
            src = a_off[i] >= 0 ? (const u16*)d.A + (unsigned)(a_off[i] + c_cb * 64 + a_chunk[i]) : zero;
          }
          regs[RS][i] = *(const u32x4*)src;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
          regs[RS][AI + i] = *(const u32x4*)b_cur[i];
          b_cur[i] += b_step[i];
        }
        if (AMODE != MVD_A_DENSE) {
          if (++c_tap == 9) {
            c_tap = 0;
            ++c_cb;
          }
#pragma unroll
          for (int i = 0; i < AI; ++i) a_off[i] = s_tab[a_tab[i] + c_tap];
        }
      };
      auto put = [&](auto rs_c, int buf) {           // this wave's granules of one k-tile: registers -> LDS slot `buf` (lane-linear 1 KiB each)
        constexpr int RS = decltype(rs_c)::value;
        unsigned char* sbase = smem + buf * STAGE + lane * 16;
#pragma unroll
        for (int i = 0; i < AI; ++i) *(u32x4*)(sbase + (lw + i * NL) * 1024) = regs[RS][i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *(u32x4*)(sbase + (A_GRAN + lw + i * NL) * 1024) = regs[RS][AI + i];
      };
      using std::integral_constant;
      // prologue: k-tiles 0 and 1 go straight to LDS, k-tiles 2 .. NBUF - 1 wait in the register stages (k-tile j in stage j % RD)
      if (0 < nkt) {
        fetch(integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        put(integral_constant<int, 0>{}, 0);
      }
      if (1 < nkt) {
        fetch(integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        put(integral_constant<int, 0>{}, 1);
      }
      unroll_steps<0, RD>([&](auto j) {
        constexpr int J = decltype(j)::value;
        if (2 + J < nkt) fetch(integral_constant<int, (2 + J) % RD>{});
      });
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                  // barrier P: k-tiles 0 and 1 written
      // iteration `it` (after B_it): write k-tile it + 2 (stage (it + 2) % RD, the OLDEST loads in flight) into slot (it + 2) % NBUF, then
      // refill that stage with k-tile it + NBUF (slot it % NBUF was released by B_it; the stage by the write just issued)
      int it = 0, wslot = 2 % NBUF;
      auto body = [&](auto rs_c, int itx) {
        asm volatile("s_barrier" ::: "memory");                                            // B_itx
        if (itx + 2 < nkt) {
          // loads in flight: k-tiles itx + 2 .. min(itx + NBUF - 1, nkt - 1): the oldest must have returned
          const int newer = (itx + NBUF - 1 < nkt ? NBUF - 1 : nkt - 1 - itx) - 2;         // stages issued after it (0 .. RD - 1)
          if (newer >= RD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((RD - 1) * LPS) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          put(rs_c, wslot);
        }
        if (itx + NBUF < nkt) fetch(rs_c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 // the writes landed before B_{itx+1}
        wslot = wslot + 1 == NBUF ? 0 : wslot + 1;
      };
      for (; it < nkt; it += RD) {
        unroll_steps<0, RD>([&](auto j) {
          constexpr int J = decltype(j)::value;
          if (it + J < nkt) body(integral_constant<int, (2 + J) % RD>{}, it + J);
        });
      }
      __syncthreads();
      return;
    }
#pragma unroll
    for (int q = 0; q < NBUF; ++q)
      if (q < nkt) stage(q);
    if (nkt >= NBUF) wait_vm_and_barrier<(NBUF - 2) * LPS>();     // barrier P: k-tiles 0 and 1 landed
    else wait_vm_and_barrier<0>();
    int bs = 0;
    for (int it = 0; it < nkt; ++it) {
      asm volatile("s_barrier" ::: "memory");                       // B_it
      if (it + NBUF < nkt) {
        stage(bs);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NBUF - 2) * LPS) : "memory");      // k-tile it+2 landed (this wave's share)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      bs = bs + 1 == NBUF ? 0 : bs + 1;
    }
    __syncthreads();
    return;
  }

  // ================================================================== CONSUMER wavefronts
#if defined(MVD_WS_VARIANT) && (MVD_WS_VARIANT & 2)
  __builtin_amdgcn_s_setprio(3);      // (probe build: the MFMA stream outranks its SIMD's loader wavefront at the issue arbiter)
#endif
  const int cm = wave / CN, cn = wave % CN;
  if (AMODE == MVD_A_DENSE && d.ln_stats != nullptr && tid < BM) {       // LayerNorm fold: {mean, rstd} of the tile's rows (BM <= 256 threads)
    const float2 st = m0 + tid < d.M ? ln_row_stats(d, m0 + tid) : make_float2(0.f, 0.f);
    s_rows[tid * 2] = st.x;
    s_rows[tid * 2 + 1] = st.y;
  }
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int fsw = (frow >> 1) & 7;
  const int fbase = (frow >> 3) * 1024 + (frow & 7) * 128;
  const int foff_hi = fbase + (((lane >> 4)) ^ fsw) * 16;
  const int foff_lo = fbase + ((4 + (lane >> 4)) ^ fsw) * 16;
  const int boff_hi = d.b_mode == MVD_B_PLANES ? foff_hi : lane * 16;
  const int boff_lo = d.b_mode == MVD_B_PLANES ? foff_lo : 1024 + lane * 16;
  auto read_frags = [&](int buf, op16x8 (&ah)[TM], op16x8 (&al)[TM], op16x8 (&bh)[TN], op16x8 (&bl)[TN]) {
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + A_GRAN * 1024;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *(const op16x8*)(sA + (cm * TM + i) * 2048 + foff_hi);
      if (NS >= 3) al[i] = *(const op16x8*)(sA + (cm * TM + i) * 2048 + foff_lo);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *(const op16x8*)(sB + (cn * TN + j) * 2048 + boff_hi);
      if (NS >= 3) bl[j] = *(const op16x8*)(sB + (cn * TN + j) * 2048 + boff_lo);
    }
  };
  auto mfma_tile = [&](const op16x8 (&ah)[TM], const op16x8 (&al)[TM], const op16x8 (&bh)[TN], const op16x8 (&bl)[TN]) {
    if (NS == 4) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(al[i], bl[j], acc[i][j], 0, 0, 0);
    }
    if (NS >= 3) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(ah[i], bl[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(ah[i], bh[j], acc[i][j], 0, 0, 0);
  };
  op16x8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
  MVD_STAMP_AT(d, wave, 1);
  wait_vm_and_barrier<0>();                                           // barrier P
  MVD_STAMP_AT(d, wave, 2);
  read_frags(0, fah[0], fal[0], fbh[0], fbl[0]);
  int br = 1 % NBUF;
  auto step = [&](auto parity, int it) {
    constexpr int Pq = decltype(parity)::value;
    wait_vm_and_barrier<0>();                                         // B_it: my reads of k-tile it returned; k-tile it+1 landed
    if (it + 1 < nkt) read_frags(br, fah[Pq ^ 1], fal[Pq ^ 1], fbh[Pq ^ 1], fbl[Pq ^ 1]);
    mfma_tile(fah[Pq], fal[Pq], fbh[Pq], fbl[Pq]);
    {
      constexpr int NM = TM * TN * NS, NR = (TM + TN) * (NS >= 3 ? 2 : 1);
      sched_reads_early<0, NR, NM>();
    }
#if !defined(MVD_WS_VARIANT) || !(MVD_WS_VARIANT & 1)
    // Keep every MFMA of a k-tile in front of the next k-tile's barrier: without this scheduling barrier the compiler sinks about half of
    // them behind it, so the two barriers of an unrolled pair of k-tiles sit 20 and 60 MFMAs apart and the loaders get 320 cycles for
    // one k-tile and 960 for the next.  Same-box A/B of the step: +1.0 % (profiles/r04_ws_variants.json; -DMVD_WS_VARIANT=1 builds without it).
    __builtin_amdgcn_sched_barrier(0);
#endif
    br = br + 1 == NBUF ? 0 : br + 1;
  };
  using std::integral_constant;
  int it = 0;
  for (; it + 2 < nkt; it += 2) {
    step(integral_constant<int, 0>{}, it);
    step(integral_constant<int, 1>{}, it + 1);
  }
  if (it < nkt) step(integral_constant<int, 0>{}, it);
  if (it + 1 < nkt) step(integral_constant<int, 1>{}, it + 1);
  MVD_STAMP_AT(d, wave, 3);
  __syncthreads();
  tile_epilogue<BM, BN, CM, CN>(p, acc, smem, m0, n0, lane, wave, AMODE == MVD_A_DENSE && d.ln_stats != nullptr ? s_rows : nullptr);
  MVD_STAMP_AT(d, wave, 8);
}

// ------------------------------------------------------------------------------------------------ 3x3 convolution, input patch in LDS
// conv_patch_kernel: stride-1 3x3 convolution whose A operand is staged ONCE per 32-channel block as the tile's input PATCH (the
// tile's pixels plus a one-pixel halo: (rows + 2) x (W + 2) pixel lines of 128 bytes) instead of nine shifted copies of the tile -- the
// nine taps of a channel block read their A fragments from the same patch at shifted pixel slots.  gemm_kernel's implicit GEMM moves
// 9 x BM pixel lines per channel block through the LDS-DMA path (all L2 hits, but the kernel is bound by what one CU can pull from
// L2 into LDS); the patch is (BM / W + 2)(W + 2) lines, 1.4 - 2.3 x BM: the A side of the operand delivery shrinks 4 - 6 x, so a
// narrow tile (128 x 80: 256 workgroups at M = 8192, N = 320 -- the whole chip) no longer pays for its low A reuse.
//   * slot p of the patch = padded pixel (segment s, patch row pr, patch column pc), p = (s (Rb + 2) + pr)(W + 2) + pc; a tile is
//     either Rb = BM / W whole rows of one image (H W >= BM) or BM / (H W) whole images (segments).  Slot p lives at byte
//     128 p of the patch buffer, its 16-byte chunk cc at position cc ^ ((p >> 1) & 7) (the DMA is lane-linear in LDS, so the
//     swizzle is applied to the per-lane SOURCE address; zero padding and rows past M source the zero page).
//   * tile row r -> centre slot c(r); tap (ky, kx) reads slot c(r) + (ky - 1)(W + 2) + (kx - 1).
//   * B: the packed weights, k order (channel block, tap) like gemm_kernel, through a ring of NB stages; two patch buffers: the
//     patch of block cb + 1 arrives in NSHARE = 11 - NB shares of PI granules per wave, issued next to the B stages of the k-tiles
//     (cb - 1, tap 8), (cb, tap 0 .. 9 - NB): after the last readers of the buffer (block cb - 1) passed their barrier, and early
//     enough that the counted vmcnt of the k-loop has retired them when (cb + 1, tap 0) is read.  Every k-tile issues the same
//     number of DMAs per wave (dummies copy the zero page into a dump granule) so the counted waits stay compile-time constants.
//   * loop: the register-pipelined ring of gemm_kernel (fragments of k-tile t + 1 read under the MFMAs of k-tile t).  Same MFMA order
//     and k order as gemm_kernel => bit-identical results.
template <int BM, int BN, int WM, int WN, int NS, int PI>
__global__ __launch_bounds__(WM * WN * 64) void conv_patch_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int B_GRAN = BN / 8, BI = (B_GRAN + NW - 1) / NW, B_GRAN_P = BI * NW;
  constexpr int BSTAGE = B_GRAN_P * 1024;
  constexpr int PSLOTS = MVD_PATCH_SLOTS_MAX, PG_MAX = PSLOTS / 8;
  constexpr int PATCH = (PG_MAX + 1) * 1024;          // + one dump granule for the dummy DMAs
  constexpr int NB = conv_patch_ring(BN, NW);
  constexpr int NSHARE = 11 - NB;
  constexpr int LPS = BI + PI;
  constexpr int PP = (PG_MAX + NW - 1) / NW;          // prologue: the whole patch of the first channel block
  constexpr int LDW = WTN + 4;
  constexpr int EPI_BYTES = NW * WTM * LDW * 4;
  constexpr int MAIN = 2 * PATCH + NB * BSTAGE;
  constexpr int SMEM = MAIN > EPI_BYTES ? MAIN : EPI_BYTES;
  static_assert(SMEM + PSLOTS * 4 <= 160 * 1024 && NB >= 2 && NSHARE >= 1, "LDS budget");
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && B_GRAN % 2 == 0, "wave tiles are made of 16x16 MFMA tiles");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM + PSLOTS * 4];

  const mvd_gemm_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int tile;
  {
    const int nb = p.tiles_n * p.tiles_m, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (p.m_fastest ? tile % p.tiles_m : tile / p.tiles_n) * BM;
  const int n0 = (p.m_fastest ? tile / p.tiles_m : tile % p.tiles_n) * BN;
  const int cb0 = blockIdx.z * (p.kt_per_split / 9);                 // kt_per_split is a multiple of 9 here (whole channel blocks)
  const int nblk = min(p.nk / 9, cb0 + p.kt_per_split / 9) - cb0;
  const int nkt = nblk * 9;

  // ---- patch geometry (uniform)
  const int W = d.Wout, H = d.Hout, HW = H * W, PW = W + 2;
  const int nseg = HW >= BM ? 1 : BM / HW, Rb = HW >= BM ? BM / W : H;
  const int seg_slots = (Rb + 2) * PW, P = nseg * seg_slots, PG = (P + 7) >> 3;
  int* s_src = (int*)(smem + SMEM);
  {
    const int b0 = m0 / HW, y0 = HW >= BM ? (m0 - b0 * HW) / W : 0;
    for (int e = tid; e < PSLOTS; e += NW * 64) {
      int off = -1;
      if (e < P) {
        const int sg = e / seg_slots, rem = e - sg * seg_slots;
        const int pr = rem / PW, pc = rem - pr * PW;
        const int b = b0 + sg, y = y0 + pr - 1, x = pc - 1;
        if (b < d.B && y >= 0 && y < H && x >= 0 && x < W) off = ((b * H + y) * W + x) * 2 * d.Cin;
      }
      s_src[e] = off;
    }
  }
  __syncthreads();
  const u16* zero = (const u16*)g_zero_page;
  unsigned char* const sB = smem + 2 * PATCH;

  int centre[TM];                      // patch slot of this lane's row of every 16-row MFMA block of the wave tile
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = (wm * TM + i) * 16 + (lane & 15);
    const int sg = r / (Rb * W), rr = r - sg * Rb * W;
    const int yy = rr / W, xx = rr - yy * W;
    centre[i] = sg * seg_slots + (yy + 1) * PW + xx + 1;
  }

  const u16* b_cur[BI];
  size_t b_step[BI];
  {
    const size_t b_kstride = (size_t)p.nt16 * 1024;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int gi = wave + i * NW;
      const int nt = (n0 >> 4) + (gi >> 1);
      const bool ok = gi < B_GRAN && nt < p.nt16;
      b_cur[i] = ok ? (const u16*)d.Wp + (size_t)nt * 1024 + (gi & 1) * 512 + lane * 8 + (size_t)cb0 * 9 * b_kstride : zero;
      b_step[i] = ok ? b_kstride : 0;
    }
  }
  auto stage_b = [&](int buf) {
    unsigned char* sbase = sB + buf * BSTAGE;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_cur[i],
                                       (__attribute__((address_space(3))) void*)(sbase + (wave + i * NW) * 1024), 16, 0, 0);
      b_cur[i] += b_step[i];
    }
  };
  // granule g of the patch of channel block `blk` (relative to cb0); not `real`: a dummy copy of the zero page into the dump granule
  auto patch_granule = [&](int g, int blk, bool real) {
    const int slot = g * 8 + (lane >> 3);
    const int off = real ? s_src[slot < PSLOTS ? slot : 0] : -1;
    const int cc = (lane & 7) ^ ((slot >> 1) & 7);
    const u16* src = off >= 0 ? (const u16*)d.A + (unsigned)(off + (cb0 + blk) * 64 + cc * 8) : zero;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + (blk & 1) * PATCH + (real ? g : PG_MAX) * 1024), 16, 0, 0);
  };
  auto patch_share = [&](int j, int blk) {
#pragma unroll
    for (int i = 0; i < PI; ++i) {
      const int g = (j * PI + i) * NW + wave;
      patch_granule(g, blk, j < NSHARE && blk < nblk && g < PG);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto mfma_tile = [&](const op16x8 (&ah)[TM], const op16x8 (&al)[TM], const op16x8 (&bh)[TN], const op16x8 (&bl)[TN]) {
    // (same term-major order as gemm_kernel: lo*lo, lo*hi, hi*lo, hi*hi per accumulator and k-tile)
    if (NS == 4) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(al[i], bl[j], acc[i][j], 0, 0, 0);
    }
    if (NS >= 3) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(ah[i], bl[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(ah[i], bh[j], acc[i][j], 0, 0, 0);
  };
  const int fg = lane >> 4;
  auto read_frags = [&](int bbuf, int pbuf, int tap, op16x8 (&ah)[TM], op16x8 (&al)[TM], op16x8 (&bh)[TN], op16x8 (&bl)[TN]) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int tapoff = (ky - 1) * PW + kx - 1;
    const unsigned char* sP = smem + pbuf * PATCH;
    const unsigned char* sBb = sB + bbuf * BSTAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int slot = centre[i] + tapoff;
      const int sw = (slot >> 1) & 7;
      ah[i] = *(const op16x8*)(sP + slot * 128 + ((fg ^ sw) << 4));
      if (NS >= 3) al[i] = *(const op16x8*)(sP + slot * 128 + (((4 + fg) ^ sw) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *(const op16x8*)(sBb + (wn * TN + j) * 2048 + lane * 16);
      if (NS >= 3) bl[j] = *(const op16x8*)(sBb + (wn * TN + j) * 2048 + 1024 + lane * 16);
    }
  };

  // ---- prologue: the whole patch of block 0, then B of k-tiles 0 .. NB-1 (each with its PI patch DMAs: the last one carries share 0
  //      of block 1, i.e. plays iteration -1; the others are dummies so that every stage is LPS DMAs)
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int g = i * NW + wave;
    patch_granule(g, 0, g < PG);
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    stage_b(q);
    patch_share(q == NB - 1 ? 0 : NSHARE, 1);
  }
  wait_vm_and_barrier<(NB - 1) * LPS>();
  op16x8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
  read_frags(0, 0, 0, fah[0], fal[0], fbh[0], fbl[0]);
  int bs = 0, br = 1 % NB, tap = 0, blk = 0;        // (tap, blk): the k-tile whose MFMAs run in the current iteration
  auto step = [&](auto parity, auto steady, int it) {
    constexpr int Pq = decltype(parity)::value;
    constexpr bool FULL = decltype(steady)::value;
    if (FULL) wait_vm_and_barrier<(NB - 2) * LPS>();
    else wait_vm_and_barrier<0>();
    const bool last_tap = tap == 8;
    if (FULL || it + NB < nkt) {
      stage_b(bs);
      patch_share(last_tap ? 0 : tap + 1, blk + (last_tap ? 2 : 1));
    }
    if (FULL || it + 1 < nkt)
      read_frags(br, (blk + (last_tap ? 1 : 0)) & 1, last_tap ? 0 : tap + 1, fah[Pq ^ 1], fal[Pq ^ 1], fbh[Pq ^ 1], fbl[Pq ^ 1]);
    mfma_tile(fah[Pq], fal[Pq], fbh[Pq], fbl[Pq]);
    if (FULL) {
      constexpr int NM = TM * TN * NS, NR = (TM + TN) * (NS >= 3 ? 2 : 1);
      sched_pattern<0, LPS + NR, NM, LPS>();
    }
    bs = bs + 1 == NB ? 0 : bs + 1;
    br = br + 1 == NB ? 0 : br + 1;
    tap = last_tap ? 0 : tap + 1;
    blk += last_tap ? 1 : 0;
  };
  using std::integral_constant;
  int it = 0;
  for (; it + NB + 1 < nkt; it += 2) {
    step(integral_constant<int, 0>{}, integral_constant<bool, true>{}, it);
    step(integral_constant<int, 1>{}, integral_constant<bool, true>{}, it + 1);
  }
  unroll_steps<0, NB + 1>([&](auto j) {
    constexpr int J = decltype(j)::value;
    if (it + J < nkt) step(integral_constant<int, J & 1>{}, integral_constant<bool, false>{}, it + J);
  });
  __syncthreads();   // the epilogue reuses the patch / stage buffers
  tile_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, lane, wave);
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
  if (d.epi == MVD_EPI_STORE) {   // 4 columns per thread: 16-byte slab reads, vector epilogue
    const size_t MN4 = MN >> 2;
    const float inv = 1.0f / d.acc_scale;   // epi_store4 re-applies acc_scale; slabs hold raw accumulators
    (void)inv;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < MN4; e += (size_t)gridDim.x * 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int z = 0; z < p.splits; ++z) {
        const float4 t = *(const float4*)(d.workspace + z * MN + e * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      const int m = (int)((e * 4) / d.N);
      const int n = (int)(e * 4 - (size_t)m * d.N);
      if (n + 3 < d.n_store) {
        epi_store4(d, m, n, v);
      } else {
        epi_store_elem(d, m, n, v.x);
        epi_store_elem(d, m, n + 1, v.y);
        epi_store_elem(d, m, n + 2, v.z);
        epi_store_elem(d, m, n + 3, v.w);
      }
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

// Split-K reduce for a GEMM whose output feeds a LayerNorm folded into its consumer (mvd_gemm_desc.rs_out): one wavefront per row and
// 256-column span; same sums and epilogue as splitk_reduce_kernel, then the row's {sum, sum of squares} over the span go to slot
// blockIdx.y of the row (wave reduction in a fixed order).  MVD_EPI_STORE, n_store == N.
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(GemmParams p) {
  const mvd_gemm_desc& d = p.d;
  const size_t MN = (size_t)d.M * d.N;
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.y * 256 + lane * 4;
  if (m >= d.M) return;
  float s1 = 0.f, q1 = 0.f;
  if (n < d.N) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* w = d.workspace + (size_t)m * d.N + n;
    for (int z = 0; z < p.splits; ++z) {
      const float4 t = *(const float4*)(w + z * MN);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float4 f = epi_store4(d, m, n, v);
    s1 = (f.x + f.y) + (f.z + f.w);
    q1 = (f.x * f.x + f.y * f.y) + (f.z * f.z + f.w * f.w);
  }
  s1 = wave_sum(s1);
  q1 = wave_sum(q1);
  if (lane == 0) *((float2*)d.rs_out + (size_t)m * d.rs_ld + blockIdx.y) = make_float2(s1, q1);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) d.rs_count[0] = gridDim.y;
}

// Split-K reduce for a GEMM whose output feeds a GroupNorm: one workgroup per (16-row slab, 256-column span); thread = (4 rows,
// one float4 column), same sums and epilogue as splitk_reduce_kernel.  The per-column {sum, sum of squares} of the slab are
// combined over the rows, then per group in column order (a group cut by the span boundary contributes from both workgroups),
// and added to the statistics of the consumer GroupNorm (gn_stats_add: integer atomics, order independent).  MVD_EPI_STORE,
// n_store == N.
template <int RPT>   // rows per thread: slabs of 4 * RPT rows (16, or 8 when the problem has too few slabs to fill the chip)
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(GemmParams p) {
  __shared__ float cs[2][4][256];
  const mvd_gemm_desc& d = p.d;
  const size_t MN = (size_t)d.M * d.N;
  const int m0 = blockIdx.x * (4 * RPT), n0 = blockIdx.y * 256;
  const int c4 = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = n0 + c4 * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < d.N) {
    float4 v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* w = d.workspace + (size_t)(m0 + rg * RPT) * d.N + n;
#pragma unroll 2
    for (int z = 0; z < p.splits; ++z) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const float4 t = *(const float4*)(w + z * MN + (size_t)r * d.N);
        v[r].x += t.x; v[r].y += t.y; v[r].z += t.z; v[r].w += t.w;
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = epi_value4(d, m0 + rg * RPT + r, n, v[r]);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const float4 f = v[r];
      epi_put4(d, m0 + rg * RPT + r, n, f);
      s.x += f.x; s.y += f.y; s.z += f.z; s.w += f.w;
      q.x += f.x * f.x; q.y += f.y * f.y; q.z += f.z * f.z; q.w += f.w * f.w;
    }
  }
  *(float4*)&cs[0][rg][c4 * 4] = s;
  *(float4*)&cs[1][rg][c4 * 4] = q;
  __syncthreads();
  const int c = threadIdx.x, nn = n0 + c;
  const float cs_ = cs[0][0][c] + cs[0][1][c] + cs[0][2][c] + cs[0][3][c];
  const float cq_ = cs[1][0][c] + cs[1][1][c] + cs[1][2][c] + cs[1][3][c];
  __syncthreads();
  cs[0][0][c] = cs_;
  cs[1][0][c] = cq_;
  __syncthreads();
  if (nn >= d.N) return;
  const int cg = d.N / d.gn_groups;
  const int g = nn / cg, pos = nn - g * cg;
  if (c != 0 && pos != 0) return;
  int len = cg - pos;
  if (len > 256 - c) len = 256 - c;
  float ss = 0.f, qq = 0.f;
  for (int j = 0; j < len; ++j) {
    ss += cs[0][0][c + j];
    qq += cs[1][0][c + j];
  }
  gn_stats_add(d.gn_stats, m0 / d.gn_hw, g, d.gn_groups, ss, qq);
}

// Split-K reduce + GroupNorm APPLY in one launch (mvd_gemm_desc.gna_out_sp): a workgroup per (image, group).  It sums the split-K slabs
// of its gn_hw x (N / groups) values, applies the STORE epilogue (scale, bias, per-image bias, residual), keeps the values in LDS, forms
// the group's mean / rstd (fp32 partial sums per thread in a fixed order, combined in double: deterministic) and writes the normalised,
// activated values as split planes -- what splitk_reduce_stats_kernel + gn_apply_stats(_cols)_kernel did in two launches with the fp32
// tensor making a round trip through memory in between (57 such pairs per configs[1] step).  The fp32 output is written unless the
// caller marks it unused; the statistics slot of the output still receives the sums (another consumer may normalise the same tensor).
// Mapping: two channels per thread (a group is N / 32 = 10, 20, 40 ... channels wide: 8-byte accesses); the 8 XCDs take runs of
// groups / 8 neighbouring groups each, so a 128-byte line of a slab row (3.2 groups of 10 channels) is pulled into one or two L2s.
#define MVD_GNK_THREADS 1024
__global__ __launch_bounds__(MVD_GNK_THREADS) void splitk_gn_kernel(GemmParams p) {
  // Latency, not bandwidth, is what this kernel has to manage: a workgroup owns a few thousand values spread over gn_hw rows, so every
  // thread takes U elements at a time and has ALL their operands -- up to ZC slabs, the residual, both biases -- in flight before it
  // touches one (a first version that walked its elements one by one was a chain of ~10 memory round trips per workgroup and slower than
  // the two kernels it replaces).  1024 threads: the largest group of a step (1024 x 10 channels) is five elements per thread.
  constexpr int NT = MVD_GNK_THREADS, NWV = NT / 64, U = 3, ZC = 4;
  extern __shared__ float s_val[];                 // [gn_hw][cg] values of the group
  __shared__ double s_red[2][NWV];
  __shared__ float s_coef[2][128];
  const mvd_gemm_desc& d = p.d;
  const int G = d.gn_groups, HW = d.gn_hw;
  // (concat mode, mvd_gemm_desc.cat_b: the GroupNorm runs over [out | cat_b], CT = N + cat_cb channels; channels >= N come from cat_b)
  const int CT = d.N + (d.cat_b ? d.cat_cb : 0);
  const int cg = CT / G, cg2 = cg >> 1;
  int g, b;
  {
    const int bid = blockIdx.x;
    if ((G & 7) == 0) {
      const int gpx = G >> 3, x = bid & 7, r = bid >> 3;
      g = x * gpx + r % gpx;
      b = r / gpx;
    } else {
      g = bid % G;
      b = bid / G;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = g * cg, m0 = b * HW;
  const size_t MN = (size_t)d.M * d.N;
  const int total = HW * cg2;
  const bool put_out = d.out != nullptr && !(d.gna_flags & MVD_GNA_OUT_UNUSED);
  // element e = (row e / cg2, channel pair e % cg2); a thread walks e = tid, tid + NT, ...: (r, j) advance without a division
  const int dq = NT / cg2, dj = NT - dq * cg2;
  const int rs = tid / cg2, js = tid - rs * cg2;                 // the thread's first element ...
  const int r0 = tid < total ? rs : 0, j0 = tid < total ? js : 0; // ... which is also its safe address for the unconditional loads
  const float* const zero2 = (const float*)g_zero_page;
  const bool has_bias = d.bias != nullptr, has_bb = d.bias_b != nullptr, has_res = d.res != nullptr;
  float s = 0.f, q = 0.f;
  {
    int r = rs, j = js;
    for (int e0 = tid; e0 < total; e0 += NT * U) {
      int rr[U], jj[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = e0 + u * NT < total;
        rr[u] = ok[u] ? r : r0;                    // (elements past the end re-read the thread's first one: every load is unconditional)
        jj[u] = ok[u] ? j : j0;
        r += dq;
        j += dj;
        if (j >= cg2) {
          j -= cg2;
          ++r;
        }
      }
      float2 acc[U], tb[U], tbb[U], tr[U];
      float2 t[ZC][U];
      const int nz0 = p.splits < ZC ? p.splits : ZC;
      bool own[U];                                  // the element is a column of THIS GEMM (else: of cat_b)
      float2 tc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = m0 + rr[u], nn = c0 + 2 * jj[u];
        own[u] = nn < d.N;
        const int n = own[u] ? nn : 0;              // (a cat_b element reads column 0 of the slabs and drops it: loads stay unconditional)
        tb[u] = *(const float2*)(has_bias ? d.bias + n : zero2);
        tbb[u] = *(const float2*)(has_bb ? d.bias_b + (size_t)(m / d.rows_per_batch) * d.ldbb + n : zero2);
        tr[u] = *(const float2*)(has_res ? d.res + (size_t)m * d.ldr + n : zero2);
        tc[u] = *(const float2*)(own[u] ? zero2 : d.cat_b + (size_t)m * d.cat_cb + (nn - d.N));
#pragma unroll
        for (int z = 0; z < ZC; ++z) {
          const float* w = d.workspace + (size_t)(z < nz0 ? z : 0) * MN + (size_t)m * d.N + n;
          t[z][u] = *(const float2*)w;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc[u] = make_float2(0.f, 0.f);
#pragma unroll
        for (int z = 0; z < ZC; ++z)
          if (z < nz0) {
            acc[u].x += t[z][u].x;
            acc[u].y += t[z][u].y;
          }
      }
      for (int zb = ZC; zb < p.splits; zb += ZC) {               // more than ZC slabs: further rounds of ZC x U loads
        const int nz = p.splits - zb < ZC ? p.splits - zb : ZC;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int m = m0 + rr[u], n = own[u] ? c0 + 2 * jj[u] : 0;
#pragma unroll
          for (int z = 0; z < ZC; ++z) t[z][u] = *(const float2*)(d.workspace + (size_t)(zb + (z < nz ? z : 0)) * MN + (size_t)m * d.N + n);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int z = 0; z < ZC; ++z)
            if (z < nz) {
              acc[u].x += t[z][u].x;
              acc[u].y += t[z][u].y;
            }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float2 v = acc[u];
        v.x = v.x * d.acc_scale + tb[u].x + tbb[u].x + tr[u].x;
        v.y = v.y * d.acc_scale + tb[u].y + tbb[u].y + tr[u].y;
        if (!own[u]) v = tc[u];
        if (ok[u]) {
          const int m = m0 + rr[u], n = c0 + 2 * jj[u];
          if (put_out && own[u]) *(float2*)(d.out + (size_t)m * d.ldo + n) = v;
          if (d.cat_raw_sp) {                      // raw planes of [out | cat_b] (the next ResBlock's 1x1 skip convolution reads them)
            uint32_t hh, ll;
            split_op16x2(v.x, v.y, hh, ll);
            u16* pp = (u16*)d.cat_raw_sp + sp_index((size_t)m, CT, n);
            *(uint32_t*)pp = hh;
            *(uint32_t*)(pp + 32) = ll;
          }
          *(float2*)(s_val + 2 * (e0 + u * NT)) = v;
          s += v.x + v.y;
          q += v.x * v.x + v.y * v.y;
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
  const double cnt = (double)HW * cg;
  const double mean = S1 / cnt;
  double var = S2 / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)d.gna_eps));
  if (tid < cg) {
    const float a = rstd * d.gna_gamma[c0 + tid];
    s_coef[0][tid] = a;
    s_coef[1][tid] = d.gna_beta[c0 + tid] - (float)mean * a;
  }
  if (tid == 0 && d.gn_stats) gn_stats_add(d.gn_stats, b, g, G, (float)S1, (float)S2);
  __syncthreads();
  u16* const ysp = (u16*)d.gna_out_sp;
  const int fl = d.gna_flags;
  {
    int r = rs, j = js;
    for (int e = tid; e < total; e += NT) {
      float2 v = *(const float2*)(s_val + 2 * e);
      v.x = v.x * s_coef[0][2 * j] + s_coef[1][2 * j];
      v.y = v.y * s_coef[0][2 * j + 1] + s_coef[1][2 * j + 1];
      if (fl & MVD_GNA_ROUND_F16) {
        v.x = (float)(_Float16)v.x;
        v.y = (float)(_Float16)v.y;
      }
      if (fl & MVD_GNA_SILU) {
        v.x = silu_f(v.x);
        v.y = silu_f(v.y);
      }
      uint32_t hh, ll;
      split_op16x2(v.x, v.y, hh, ll);
      u16* pp = ysp + sp_index((size_t)(m0 + r), CT, c0 + 2 * j);
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

// Tile configurations (mvd_gemm_desc.cfg = 1 + MVD_GEMM_CFG_STRIDE * tile + 2 * loop + order; 0 = built-in heuristic).
//   tile : 0 = 64x64 (2x2 waves)  1 = 128x128 (2x4)  2 = 128x80 (4x1)  3 = 64x80 (4x1)  4 = 128x160 (4x2)
//          (a 256x128 tile -- 128x32 wave tiles, 64 MFMAs per k-tile and wave against 20 fragment reads and 6 DMAs -- was built and
//          measured in round 3: equal or slower on every shape of the step, profiles/r03_gemm_tile256_probe.log; dropped)
//   loop : 0 = plain two-buffer loop, 1 = register-pipelined loop, 2 = staggered wave groups, 3 LDS buffers (8-wave tiles 1 and
//          4 only), 3 = staggered, 4 LDS buffers (tile 1 only: 128 KiB), 4 = register-pipelined loop over a ring of <= 4 LDS buffers,
//          5 = over a ring of <= 8 (4-wave tiles 0, 2, 3 only: the 8-wave tiles fit 4), 6 = conv_patch_kernel (stride-1 3x3 convolutions, tiles 1, 2, 4),
//          7 = gemm_ws_kernel (consumer / loader wavefronts, LDS-DMA delivery), 8 = gemm_ws_kernel with register-staged delivery (LM = 1),
//          9 = gemm_kernel with register-staged delivery (global_load -> VGPR -> ds_write_b128, two LDS buffers; tiles 0 - 3)
//   order : 0 = n-fastest tile order, 1 = m-fastest
// The 80-column family serves MVD_EPI_STORE only (the GEGLU / QKV epilogues walk a wave tile in 32-column blocks).
struct TileInfo {
  int bm, bn, waves;
  int cores_plain, cores_pipe;   // workgroups that fit one CU (LDS / registers), per loop variant
};
static const TileInfo kTiles[MVD_GEMM_TILES] = {{64, 64, 4, 5, 3}, {128, 128, 8, 2, 1}, {128, 80, 4, 2, 2}, {64, 80, 4, 4, 3},
                                                {128, 160, 8, 1, 1}};

// Split-K selection by a small time model (unit: 0.7 us ~ one DMA round trip).  What matters most is how evenly
// tiles*splits workgroups divide over the 256 CUs (192 tiles: 1, 2 or 3 splits all leave a CU with 180 k-tiles, 4 splits
// give every CU 3 x 45), then whether enough workgroups are co-resident to hide the per-k-tile DMA latency, then the cost
// of the fp32 partial-sum round trip.
static int choose_splits(long tiles, int nk, const TileInfo& ti, int loop, size_t mn) {
  const double area = (double)ti.bm * ti.bn / (128.0 * 128.0);
  const double t_mfma = 0.75 * area;                 // MFMA-pipe time of one k-tile of one workgroup
  const double t_lat = loop == 0 ? 1.0 : (loop == 1 ? 0.4 : (loop == 2 ? 0.2 : (loop == 5 ? 0.05 : 0.1)));      // (loops 3, 4, 6: 0.1)   // exposed DMA latency per k-tile, workgroup alone
  const double t_epi = 0.27 + 1.73 * area;
  const int coresident = loop == 0 ? ti.cores_plain : (loop == 1 ? ti.cores_pipe : (loop == 4 && ti.waves == 4 ? 2 : 1));      // (loop 7: 1)
  const double red_fixed = 6.0, red_per_split = (double)mn * 8.0 / 3.0e12 / 0.7e-6;
  int best = 1;
  double best_t = 1e30;
  const int smax = nk / 8 < 1 ? 1 : (nk / 8 > 32 ? 32 : nk / 8);
  for (int sp = 1; sp <= smax; ++sp) {
    const int iters = (nk + sp - 1) / sp;
    if ((long)iters * (sp - 1) >= nk) continue;      // an empty trailing split
    const long per_cu = (tiles * sp + 255) / 256;
    const long rounds = (per_cu + coresident - 1) / coresident;
    const double busy = (double)per_cu * (iters * t_mfma + t_epi);
    const double lat = (double)rounds * (iters * t_lat + t_epi);
    double t = busy > lat ? busy : lat;
    if (sp > 1) t += red_fixed + sp * red_per_split;
    if (t < best_t * 0.97) {                          // prefer fewer splits unless clearly better
      best_t = t;
      best = sp;
    }
  }
  return best;
}

template <int BM, int BN, int WM, int WN, int STAGES>
void launch_cfg(GemmParams& p, hipStream_t s) {
  dim3 grid(p.tiles_n * p.tiles_m, 1, p.splits), block(WM * WN * 64);
  const bool conv = p.d.a_mode == MVD_A_CONV3X3;
  const int ns = p.d.prec;
  if (!conv && ns == 4) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 4, MVD_A_DENSE, STAGES>), grid, block, 0, s, p);
  if (!conv && ns == 3) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 3, MVD_A_DENSE, STAGES>), grid, block, 0, s, p);
  if (!conv && ns == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, MVD_A_DENSE, STAGES>), grid, block, 0, s, p);
  if (conv && ns == 4) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 4, MVD_A_CONV3X3, STAGES>), grid, block, 0, s, p);
  if (conv && ns == 3) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 3, MVD_A_CONV3X3, STAGES>), grid, block, 0, s, p);
  if (conv && ns == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, MVD_A_CONV3X3, STAGES>), grid, block, 0, s, p);
}

template <int BM, int BN, int CM, int CN, int LM = 0>
void launch_ws(GemmParams& p, hipStream_t s) {
  dim3 grid(p.tiles_n * p.tiles_m, 1, p.splits), block(512);
  const bool conv = p.d.a_mode == MVD_A_CONV3X3;
  const int ns = p.d.prec;
  if (!conv && ns == 4) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 4, MVD_A_DENSE, LM>), grid, block, 0, s, p);
  if (!conv && ns == 3) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 3, MVD_A_DENSE, LM>), grid, block, 0, s, p);
  if (!conv && ns == 1) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 1, MVD_A_DENSE, LM>), grid, block, 0, s, p);
  if (conv && ns == 4) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 4, MVD_A_CONV3X3, LM>), grid, block, 0, s, p);
  if (conv && ns == 3) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 3, MVD_A_CONV3X3, LM>), grid, block, 0, s, p);
  if (conv && ns == 1) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 1, MVD_A_CONV3X3, LM>), grid, block, 0, s, p);
}

template <int BM, int BN, int WM, int WN>
void launch_patch(GemmParams& p, hipStream_t s, int pi) {
  dim3 grid(p.tiles_n * p.tiles_m, 1, p.splits), block(WM * WN * 64);
  const int ns = p.d.prec;
#define MVD_PATCH_CASE(NS_, PI_) \
  if (ns == NS_ && pi == PI_) hipLaunchKernelGGL((conv_patch_kernel<BM, BN, WM, WN, NS_, PI_>), grid, block, 0, s, p);
  MVD_PATCH_CASE(4, 1) MVD_PATCH_CASE(4, 2) MVD_PATCH_CASE(3, 1) MVD_PATCH_CASE(3, 2) MVD_PATCH_CASE(1, 1) MVD_PATCH_CASE(1, 2)
#undef MVD_PATCH_CASE
}

// conv_patch_kernel serves stride-1, padded 3x3 convolutions whose BM-row tiles are whole image rows of one image or whole images;
// returns the patch DMAs per wave and k-tile (1 or 2), 0 when the problem does not fit.
static int patch_shares(const mvd_gemm_desc& d, const TileInfo& ti) {
  if (d.a_mode != MVD_A_CONV3X3 || d.b_mode != MVD_B_PACKED || d.stride != 1 || d.upsample || d.no_pad_tl) return 0;
  if (d.Hin != d.Hout || d.Win != d.Wout || ti.bm != 128) return 0;
  const int W = d.Wout, HW = d.Hout * d.Wout;
  if (ti.bm % W != 0 || (HW >= ti.bm ? HW % ti.bm != 0 : ti.bm % HW != 0)) return 0;
  const int nseg = HW >= ti.bm ? 1 : ti.bm / HW, Rb = HW >= ti.bm ? ti.bm / W : d.Hout;
  const int P = nseg * (Rb + 2) * (W + 2);
  if (P > MVD_PATCH_SLOTS_MAX) return 0;
  const int PG = (P + 7) / 8, cap = (11 - conv_patch_ring(ti.bn, ti.waves)) * ti.waves;
  return PG <= cap ? 1 : (PG <= 2 * cap ? 2 : 0);
}

static bool cfg_supported(const mvd_gemm_desc& d, int cfg) {
  if (cfg == 0) return true;
  if (cfg < 0 || cfg > MVD_GEMM_CFG_STRIDE * MVD_GEMM_TILES) return false;
  const int tile = (cfg - 1) / MVD_GEMM_CFG_STRIDE, loop = ((cfg - 1) % MVD_GEMM_CFG_STRIDE) >> 1;
  if (loop >= MVD_GEMM_LOOPS) return false;
  if (tile >= 2 && d.epi != MVD_EPI_STORE) return false;
  const int waves = kTiles[tile].waves;
  if ((loop == 2 || loop == 3) && waves != 8) return false;
  if (loop == 3 && tile != 1) return false;
  if (loop == 5 && waves != 4) return false;
  if (loop == 6) return (tile == 1 || tile == 2 || tile == 4) && patch_shares(d, kTiles[tile]) > 0;
  if (loop == 7 || loop == 8) return tile == 1 || ((tile == 2 || tile == 4) && d.epi == MVD_EPI_STORE);   // (64x64 wave tiles: every epilogue)
  if (loop == 9) return tile <= 3;
  if (loop == 10) return tile == 1 && mvd_gemm_pt_supported(d);      // gemm_pt.hip: the persistent role-split kernel (128x128 tiles, every epilogue)
  return true;
}

}  // namespace

extern "C" int mvd_gemm_cfg_supported(const mvd_gemm_desc* dp, int cfg) { return dp && cfg_supported(*dp, cfg) ? 1 : 0; }

extern "C" int mvd_gemm(const mvd_gemm_desc* dp, mvd_stream_t stream) {
  MVD_CHECK_ARG(dp != nullptr, "mvd_gemm: null descriptor");
  GemmParams p;
  p.d = *dp;
  mvd_gemm_desc& d = p.d;
  MVD_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "mvd_gemm: bad sizes M=%d N=%d K=%d", d.M, d.N, d.K);
  MVD_CHECK_ARG(d.K % 32 == 0, "mvd_gemm: K=%d must be a multiple of 32 (pad the packed weight)", d.K);
  MVD_CHECK_ARG(d.N % 16 == 0, "mvd_gemm: N=%d must be a multiple of 16 (pad the packed weight)", d.N);
  MVD_CHECK_ARG(d.prec == MVD_PREC_X1 || d.prec == MVD_PREC_X3 || d.prec == MVD_PREC_X4, "mvd_gemm: bad prec %d", d.prec);
  MVD_CHECK_ARG(d.A && d.Wp, "mvd_gemm: null operand");
  MVD_CHECK_ARG(((uintptr_t)d.A & 127) == 0 && ((uintptr_t)d.Wp & 127) == 0, "mvd_gemm: operands must be 128-byte aligned");
  if (d.a_mode == MVD_A_CONV3X3) {
    MVD_CHECK_ARG(d.Cin % 32 == 0 && d.K == 9 * d.Cin, "mvd_gemm: conv needs Cin %% 32 == 0 and K == 9*Cin (Cin=%d K=%d)", d.Cin, d.K);
    MVD_CHECK_ARG(d.M == d.B * d.Hout * d.Wout, "mvd_gemm: conv M mismatch");
    MVD_CHECK_ARG(d.stride == 1 || d.stride == 2, "mvd_gemm: conv stride must be 1 or 2");
    if (d.upsample) MVD_CHECK_ARG(d.stride == 1 && d.Hout == 2 * d.Hin && d.Wout == 2 * d.Win, "mvd_gemm: upsample geometry");
  } else {
    MVD_CHECK_ARG(d.a_mode == MVD_A_DENSE, "mvd_gemm: bad a_mode");
    MVD_CHECK_ARG(d.lda >= d.K && d.lda % 32 == 0, "mvd_gemm: lda=%d must be >= K=%d and a multiple of 32", d.lda, d.K);
  }
  if (d.out_sp) MVD_CHECK_ARG(d.ldp > 0 && d.ldp % 32 == 0 && ((uintptr_t)d.out_sp & 127) == 0,
                              "mvd_gemm: split-plane output needs ldp %% 32 == 0 and 128-byte alignment");
  if (d.out) MVD_CHECK_ARG(d.ldo % 4 == 0 && ((uintptr_t)d.out & 15) == 0, "mvd_gemm: out must be 16-byte aligned with ldo %% 4 == 0");
  if (d.res) MVD_CHECK_ARG(d.ldr % 4 == 0 && ((uintptr_t)d.res & 15) == 0, "mvd_gemm: res must be 16-byte aligned with ldr %% 4 == 0");
  MVD_CHECK_ARG(((uintptr_t)d.bias & 15) == 0 && ((uintptr_t)d.bias_b & 15) == 0 && ((uintptr_t)d.colscale & 15) == 0,
                "mvd_gemm: bias / bias_b / colscale must be 16-byte aligned");
  if (d.epi == MVD_EPI_STORE) {
    MVD_CHECK_ARG(d.out != nullptr || d.out_sp != nullptr, "mvd_gemm: no output");
    if (d.n_store <= 0 || d.n_store > d.N) d.n_store = d.N;
    if (d.bias_b) {
      MVD_CHECK_ARG(d.rows_per_batch > 0, "mvd_gemm: bias_b needs rows_per_batch");
      if (d.ldbb == 0) d.ldbb = d.N;
      MVD_CHECK_ARG(d.ldbb % 4 == 0, "mvd_gemm: ldbb %% 4 != 0");
    }
  } else if (d.epi == MVD_EPI_GEGLU) {
    MVD_CHECK_ARG((d.out != nullptr || d.out_sp != nullptr) && d.N % 32 == 0, "mvd_gemm: GEGLU needs an output and N %% 32 == 0");
    d.bias_b = nullptr;
  } else if (d.epi == MVD_EPI_QKV) {
    MVD_CHECK_ARG(d.q_hi && d.q_lo && d.k_hi && d.k_lo && d.vt_hi && d.vt_lo, "mvd_gemm: QKV planes missing");
    MVD_CHECK_ARG(d.heads > 0 && d.dhead > 0 && d.N == 3 * d.heads * d.dhead, "mvd_gemm: QKV needs N == 3*heads*dhead");
    MVD_CHECK_ARG(d.L > 0 && d.M % d.L == 0 && d.Lpad >= d.L, "mvd_gemm: QKV needs M %% L == 0");
    MVD_CHECK_ARG(d.dhead % 4 == 0 && d.L % 4 == 0, "mvd_gemm: QKV needs dhead %% 4 == 0 and L %% 4 == 0");
  } else {
    MVD_CHECK_ARG(false, "mvd_gemm: bad epilogue %d", d.epi);
  }
  if (d.rs_out)
    MVD_CHECK_ARG(d.epi == MVD_EPI_STORE && d.n_store == d.N && !d.gn_stats && d.rs_count && d.rs_ld >= cdiv(d.N, 32) && ((uintptr_t)d.rs_out & 7) == 0,
                  "mvd_gemm: rs_out needs MVD_EPI_STORE, n_store == N, no gn_stats, rs_count and rs_ld >= N / 32 (N=%d rs_ld=%d)", d.N, d.rs_ld);
  if (d.ln_stats) {
    MVD_CHECK_ARG((d.epi == MVD_EPI_QKV || d.epi == MVD_EPI_GEGLU) && d.ln_count && d.ln_colsum && d.ln_dim > 0 && d.ln_ld > 0 &&
                      ((uintptr_t)d.ln_stats & 7) == 0 && ((uintptr_t)d.ln_colsum & 15) == 0,
                  "mvd_gemm: ln_stats (LayerNorm fold) serves the QKV / GEGLU epilogues and needs ln_count, ln_colsum, ln_dim, ln_ld");
    d.splitk = 1;      // the fold lives in the tile epilogue (the row statistics are per tile row)
  }
  if (d.gn_stats)
    MVD_CHECK_ARG(d.epi == MVD_EPI_STORE && d.n_store == d.N && d.M % 16 == 0 && d.gn_hw > 0 && d.gn_hw % 16 == 0 && d.gn_groups > 0 &&
                      d.N % d.gn_groups == 0,
                  "mvd_gemm: gn_stats needs MVD_EPI_STORE, n_store == N, M %% 16 == 0, gn_hw %% 16 == 0, N %% gn_groups == 0 (N=%d M=%d hw=%d)",
                  d.N, d.M, d.gn_hw);
  if (d.gna_out_sp)
    MVD_CHECK_ARG(d.gn_stats && d.out && d.ldo == d.N && d.gna_gamma && d.gna_beta && d.N % 32 == 0 && d.M % d.gn_hw == 0 &&
                      ((uintptr_t)d.gna_out_sp & 127) == 0,
                  "mvd_gemm: gna_out_sp (GroupNorm apply behind the GEMM) needs gn_stats, out with ldo == N, gamma / beta, N %% 32 == 0, M %% gn_hw == 0");
  if (d.cat_b)
    MVD_CHECK_ARG(d.gna_out_sp && d.cat_cb > 0 && d.cat_cb % 2 == 0 && (d.N + d.cat_cb) % 32 == 0 && (d.N + d.cat_cb) % d.gn_groups == 0 &&
                      ((uintptr_t)d.cat_b & 7) == 0 && ((uintptr_t)d.cat_raw_sp & 127) == 0 &&
                      mvd_concat_groupnorm_fits(d.N, d.cat_cb, d.gn_hw, d.gn_groups),
                  "mvd_gemm: cat_b (GroupNorm over [out | cat_b]) needs gna_out_sp, an even cat_cb, (N + cat_cb) %% 32 == 0 and a shape "
                  "mvd_concat_groupnorm_fits() accepts (N=%d cat_cb=%d hw=%d)", d.N, d.cat_cb, d.gn_hw);
  long long* const gna_stats = d.gn_stats;       // the statistics slot of the tensor the GroupNorm normalises ...
  if (d.cat_b) d.gn_stats = nullptr;             // ... which in concat mode is [out | cat_b]: the GEMM's own epilogue / reduce must not touch it
  if (d.b_mode == MVD_B_PLANES)
    MVD_CHECK_ARG(d.ldb >= d.K && d.ldb % 32 == 0, "mvd_gemm: B planes need ldb=%d >= K=%d, a multiple of 32", d.ldb, d.K);
  else
    MVD_CHECK_ARG(d.b_mode == MVD_B_PACKED, "mvd_gemm: bad b_mode %d", d.b_mode);
  if (d.acc_scale == 0.f) d.acc_scale = 1.f;
  p.nk = d.K / 32;
  p.nt16 = d.N / 16;
  // ---- kernel configuration: explicit (cfg >= 1) or the built-in heuristic (128x128 once the grid fills the chip, else 64x64)
  int tile, loop = 0, order = -1;
  MVD_CHECK_ARG(d.cfg >= 0 && d.cfg <= MVD_GEMM_CFG_STRIDE * MVD_GEMM_TILES, "mvd_gemm: bad cfg %d", d.cfg);
  if (d.cfg >= 1) {
    tile = (d.cfg - 1) / MVD_GEMM_CFG_STRIDE;
    loop = ((d.cfg - 1) % MVD_GEMM_CFG_STRIDE) >> 1;
    order = (d.cfg - 1) & 1;
    if (loop == 10 && tile == 1 && !mvd_gemm_pt_supported(d)) loop = 0;   // (a problem the persistent kernel does not take -- K < 64, ragged n_store --
                                                                            //  runs the plain loop of the same tile; mvd_gemm_cfg_supported says so)
    else
    MVD_CHECK_ARG(cfg_supported(d, d.cfg), "mvd_gemm: cfg %d (tile %d, loop %d) does not serve this problem (include/mvd_hip.h: cfg)", d.cfg, tile,
                  loop);
  } else {
    const long tiles128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    tile = (tiles128 >= 128 && (d.N >= 512 || d.K >= 2048)) ? 1 : 0;
  }
  const TileInfo& ti = kTiles[tile];
  p.tiles_n = cdiv(d.N, ti.bn);
  p.tiles_m = cdiv(d.M, ti.bm);
  if (order < 0) {
    // bytes each XCD pulls through its L2 under the two tile orders (8 XCDs, operands are 4 B per element)
    const double W = (double)d.N * d.K * 4.0;
    const double A = (double)d.M * d.K * 4.0 / (d.a_mode == MVD_A_CONV3X3 ? 9.0 : 1.0);
    const double rep_m = p.tiles_m < 8 ? p.tiles_m : 8, rep_n = p.tiles_n < 8 ? p.tiles_n : 8;
    order = (W + A * rep_n) < (W * rep_m + A) ? 1 : 0;
  }
  p.m_fastest = order;
  int splits = d.splitk;
  const long tiles = (long)p.tiles_m * p.tiles_n;
  if (splits == 0) splits = choose_splits(tiles, p.nk, ti, loop, (size_t)d.M * d.N);
  if (splits < 1) splits = 1;
  if (splits > p.nk) splits = p.nk;
  if (splits > 1) {
    if (d.workspace == nullptr) splits = 1;
    else {
      const size_t cap = d.workspace_elems / ((size_t)d.M * d.N);
      if ((size_t)splits > cap) splits = cap < 1 ? 1 : (int)cap;
    }
  }
  if (loop == 10) {       // gemm_pt_kernel: every split needs >= 2 k-tiles (a tile's last two ring slots become its staging tile)
    const int need = mvd_gemm_pt_min_ktiles();
    while (splits > 1) {
      const int kps = cdiv(p.nk, splits), ns = cdiv(p.nk, kps);
      if (p.nk - (ns - 1) * kps >= need) break;
      --splits;
    }
  }
  p.kt_per_split = cdiv(p.nk, splits);
  if (loop == 6) p.kt_per_split = 9 * cdiv(p.nk / 9, splits);       // conv_patch_kernel: a split is a run of whole channel blocks
  p.splits = cdiv(p.nk, p.kt_per_split);
  hipStream_t s = (hipStream_t)stream;
  switch (tile * 16 + loop) {
    case 0: launch_cfg<64, 64, 2, 2, 2>(p, s); break;
    case 1: launch_cfg<64, 64, 2, 2, 3>(p, s); break;
    case 4: launch_cfg<64, 64, 2, 2, 6>(p, s); break;
    case 5: launch_cfg<64, 64, 2, 2, 7>(p, s); break;
    case 16: launch_cfg<128, 128, 2, 4, 2>(p, s); break;
    case 17: launch_cfg<128, 128, 2, 4, 3>(p, s); break;
    case 18: launch_cfg<128, 128, 2, 4, 4>(p, s); break;
    case 19: launch_cfg<128, 128, 2, 4, 5>(p, s); break;
    case 20: launch_cfg<128, 128, 2, 4, 6>(p, s); break;
    case 32: launch_cfg<128, 80, 4, 1, 2>(p, s); break;
    case 33: launch_cfg<128, 80, 4, 1, 3>(p, s); break;
    case 36: launch_cfg<128, 80, 4, 1, 6>(p, s); break;
    case 37: launch_cfg<128, 80, 4, 1, 7>(p, s); break;
    case 48: launch_cfg<64, 80, 4, 1, 2>(p, s); break;
    case 49: launch_cfg<64, 80, 4, 1, 3>(p, s); break;
    case 52: launch_cfg<64, 80, 4, 1, 6>(p, s); break;
    case 53: launch_cfg<64, 80, 4, 1, 7>(p, s); break;
    case 64: launch_cfg<128, 160, 4, 2, 2>(p, s); break;
    case 65: launch_cfg<128, 160, 4, 2, 3>(p, s); break;
    case 66: launch_cfg<128, 160, 4, 2, 4>(p, s); break;
    case 68: launch_cfg<128, 160, 4, 2, 6>(p, s); break;
    case 22: launch_patch<128, 128, 2, 4>(p, s, patch_shares(d, ti)); break;
    case 38: launch_patch<128, 80, 4, 1>(p, s, patch_shares(d, ti)); break;
    case 70: launch_patch<128, 160, 4, 2>(p, s, patch_shares(d, ti)); break;
    case 23: launch_ws<128, 128, 2, 2>(p, s); break;
    case 39: launch_ws<128, 80, 4, 1>(p, s); break;
    case 71: launch_ws<128, 160, 2, 2>(p, s); break;
    case 9: launch_cfg<64, 64, 2, 2, 8>(p, s); break;
    case 41: launch_cfg<128, 80, 4, 1, 8>(p, s); break;
    case 57: launch_cfg<64, 80, 4, 1, 8>(p, s); break;
    case 25: launch_cfg<128, 128, 2, 4, 8>(p, s); break;
    case 24: launch_ws<128, 128, 2, 2, 1>(p, s); break;
    case 40: launch_ws<128, 80, 4, 1, 1>(p, s); break;
    case 72: launch_ws<128, 160, 2, 2, 1>(p, s); break;
    case 26: mvd_gemm_pt_launch(p, s); break;
    default: MVD_CHECK_ARG(false, "mvd_gemm: no kernel for tile %d loop %d", tile, loop);
  }
  MVD_CHECK_LAUNCH("mvd_gemm");
  // GroupNorm apply behind the GEMM (gna_out_sp): one reduce + apply kernel when the (image, group) slab of a split GEMM fits the LDS,
  // else the ordinary producer statistics followed by the apply kernel
  const int gna_ct = d.N + (d.cat_b ? d.cat_cb : 0);
  const int gna_cg = d.gna_out_sp ? gna_ct / d.gn_groups : 0;
  const size_t gna_lds = (size_t)d.gn_hw * gna_cg * 4;
  const bool gna_fused = d.gna_out_sp && p.splits > 1 && !d.rs_out && !d.out_sp && !d.colscale && d.act == MVD_ACT_NONE && (gna_cg & 1) == 0 &&
                         gna_cg <= 128 && gna_lds <= 128 * 1024 && d.ldo % 2 == 0 && (!d.res || d.ldr % 2 == 0) && (!d.bias_b || d.ldbb % 2 == 0) &&
                         d.N % 2 == 0;
  if (gna_fused) {
    {                  // (more than the default 64 KiB of dynamic LDS: 1024 rows x 30 channels of a concatenation)
      static unsigned long long raised = 0;
      const hipError_t e = mvd_raise_dynamic_lds((const void*)splitk_gn_kernel, 128 * 1024, &raised);
      MVD_CHECK_ARG(e == hipSuccess, "mvd_gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
    }
    p.d.gn_stats = gna_stats;      // (the slot of the normalised tensor -- of the concatenation in concat mode)
    hipLaunchKernelGGL(splitk_gn_kernel, dim3((d.M / d.gn_hw) * d.gn_groups), dim3(MVD_GNK_THREADS), gna_lds, s, p);
    MVD_CHECK_LAUNCH("mvd_gemm/splitk_gn");
    return 0;
  }
  if (p.splits > 1) {
    if (d.rs_out) {
      hipLaunchKernelGGL(splitk_reduce_rows_kernel, dim3(cdiv(d.M, 4), cdiv(d.N, 256)), dim3(256), 0, s, p);
    } else if (d.gn_stats) {
      const int spans = (d.N + 255) / 256;
      if ((d.M / 16) * spans >= 1024)
        hipLaunchKernelGGL(splitk_reduce_stats_kernel<4>, dim3(d.M / 16, spans), dim3(256), 0, s, p);
      else
        hipLaunchKernelGGL(splitk_reduce_stats_kernel<2>, dim3(d.M / 8, spans), dim3(256), 0, s, p);
    } else {
      const size_t total = (size_t)d.M * d.N;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
    }
    MVD_CHECK_LAUNCH("mvd_gemm/splitk_reduce");
  }
  if (d.gna_out_sp && d.cat_b)     // concat mode without the fused reduce: the output just written + cat_b -> one concat-and-normalise launch
    return mvd_concat_groupnorm(d.out, d.N, d.cat_b, d.cat_cb, nullptr, d.cat_raw_sp, d.gna_out_sp, d.gna_gamma, d.gna_beta, gna_stats,
                                d.M / d.gn_hw, d.gn_hw, d.gn_groups, d.gna_eps, d.gna_flags & (MVD_GNA_SILU | MVD_GNA_ROUND_F16), stream);
  if (d.gna_out_sp)
    return mvd_groupnorm_from_stats(d.out, d.gna_out_sp, d.gna_gamma, d.gna_beta, d.gn_stats, d.M / d.gn_hw, d.gn_hw, d.N, d.gn_groups,
                                    d.gna_eps, d.gna_flags & (MVD_GNA_SILU | MVD_GNA_ROUND_F16), stream);
  return 0;
}

// ------------------------------------------------------------------------------------------------ packing
namespace {
// one thread per packed element pair (hi, lo)
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, u16* __restrict__ out, int N, int K,
                                                   int Np, int Kp, int ldw, int geglu, int conv_cin, int conv_cin_pad,
                                                   float scale) {
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
        const int blk = k >> 5;                      // K order: (32-channel block, tap, channel in block)
        const int cb = blk / 9, tap = blk - cb * 9;
        const int ci = cb * 32 + (k & 31);
        if (ci < conv_cin) v = w[((size_t)src_n * conv_cin + ci) * 9 + tap];
      } else if (k < K) {
        v = w[(size_t)src_n * ldw + k];
      }
    }
    u16 hi, lo;
    split_op16(v * scale, hi, lo);
    // [kt][nt][hi image | lo image]; an image is the 16x16x32 MFMA B fragment of the micro-tile as the wave holds it: lane
    // l = (n & 15) + 16 * (k-chunk of 8) owns 16 contiguous bytes -- one fully coalesced 1 KiB access per image
    // (the LDS-DMA source of a granule is contiguous, and the fragment reads from LDS are lane-contiguous: no bank conflicts)
    const int kt = k >> 5, kk = k & 31, nt = n >> 4, nn = n & 15;
    const size_t base = ((size_t)kt * (Np >> 4) + nt) * 1024 + (nn + 16 * (kk >> 3)) * 8 + (kk & 7);
    out[base] = hi;
    out[base + 512] = lo;
  }
}

// fp32 (rows, cols) with leading dim ldx -> split planes (rows, ldp); columns [cols, ldp) are zero filled
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, u16* __restrict__ sp, size_t rows, int cols,
                                                           int ldx, int ldp) {
  const int c4 = ldp >> 2;
  const size_t total = rows * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / c4;
    const int c = (int)(e - r * c4) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (c + j) < cols ? x[r * ldx + c + j] : 0.f;
    store_sp4(sp, r, ldp, c, v[0], v[1], v[2], v[3]);
  }
}
}  // namespace

extern "C" size_t mvd_packed_weight_bytes(int N, int K) {
  const size_t Np = (size_t)((N + 15) & ~15), Kp = (size_t)((K + 31) & ~31);
  return Np * Kp * 4;
}

extern "C" int mvd_operand_format(void) { return MVD_OPERAND_FORMAT; }

extern "C" int mvd_pack_linear_weight(const float* w, int N, int K, int ldw, int geglu, float scale, void* packed,
                                      mvd_stream_t stream) {
  MVD_CHECK_ARG(w && packed && N > 0 && K > 0 && ldw >= K, "mvd_pack_linear_weight: bad arguments");
  if (geglu) MVD_CHECK_ARG(N % 32 == 0, "mvd_pack_linear_weight: geglu needs N %% 32 == 0");
  const int Np = (N + 15) & ~15, Kp = (K + 31) & ~31;
  const size_t total = (size_t)Np * Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (u16*)packed, N, K, Np, Kp, ldw,
                     geglu, 0, 0, scale);
  MVD_CHECK_LAUNCH("mvd_pack_linear_weight");
  return 0;
}

extern "C" int mvd_pack_conv3x3_weight(const float* w, int Cout, int Cin, int cin_pad, float scale, void* packed,
                                       mvd_stream_t stream) {
  MVD_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && cin_pad >= Cin && cin_pad % 32 == 0,
                "mvd_pack_conv3x3_weight: bad arguments (cin_pad must be a multiple of 32)");
  const int Np = (Cout + 15) & ~15, Kp = 9 * cin_pad;
  const size_t total = (size_t)Np * Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (u16*)packed, Cout, Kp, Np, Kp, 0,
                     0, Cin, cin_pad, scale);
  MVD_CHECK_LAUNCH("mvd_pack_conv3x3_weight");
  return 0;
}

extern "C" int mvd_split_planes(const float* x, void* sp, size_t rows, int cols, int ldx, int ldp, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && sp && rows > 0 && cols > 0 && ldx >= cols && ldp >= cols && ldp % 32 == 0,
                "mvd_split_planes: bad arguments (ldp must be a multiple of 32)");
  const size_t total = rows * (size_t)(ldp / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (u16*)sp, rows, cols, ldx, ldp);
  MVD_CHECK_LAUNCH("mvd_split_planes");
  return 0;
}
