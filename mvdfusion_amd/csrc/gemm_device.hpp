// Device code shared by the GEMM kernels (gemm_plain.hpp, gemm_ws.hip, gemm_patch.hip) and the split-K reduce kernels (gemm.hip): the zero
// page, the per-element epilogue helpers, the tile epilogue (STORE / GEGLU / QKV with their fused statistics) and the wait / scheduling
// helpers of the k loops.  Header-only in an unnamed namespace: every translation unit of the GEMM family gets its own copy, so one
// kernel family can be edited and rebuilt without recompiling the others (mvdfusion_amd/csrc/build.py compiles them in parallel).
#pragma once
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
  v *= gemm_acc_scale(d);
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
  v *= gemm_acc_scale(d);
  g *= gemm_acc_scale(d);
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
  v.x *= gemm_acc_scale(d); v.y *= gemm_acc_scale(d); v.z *= gemm_acc_scale(d); v.w *= gemm_acc_scale(d);
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
    const float scale = gemm_acc_scale(d);
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
      const float scale = gemm_acc_scale(d);
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
      const float scale = gemm_acc_scale(d);
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
    const float scale = gemm_acc_scale(d);
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
// (consumer wavefronts of gemm_ws_kernel: they own no DMA; their prefetch requests must not be waited for)
__device__ __forceinline__ void wait_lgkm_and_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

}  // namespace
