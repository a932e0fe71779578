// Attention kernels.
//  * attn_kernel      : flash-style self-attention over the tokens of one view on bf16 MFMA (16x16x32), operands are
//                       the pre-split bf16 planes written by the QKV GEMM epilogue (gemm.hip, MVD_EPI_QKV).
//                       Computes S^T = K Q^T so that every lane owns ONE query column: the online-softmax row
//                       statistics need only two cross-lane shuffles, and P^T feeds the PV MFMA's B operand straight
//                       from registers (the MFMA k-slot order is permuted identically on the V^T side).
//  * pixel_xattn_kernel : 1 query x D context tokens per pixel (DualAttnetionBlock attn2), VALU.
//  * view_mha_kernel  : timm Attention core over the V reference views of GridAttn (sequence length V <= 16), VALU.
//  * view_pool_kernel : weight_layer + softmax over V + weighted sum.
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

constexpr int KV_TILE = 64;

// QT = 16-query tiles per wavefront (1 or 2).  A workgroup = 4 wavefronts = 64 * QT queries of one (batch, head).  K / V^T tiles
// of 64 keys go through LDS with issue-early / write-late register staging: the global loads of key tile t+1 are issued before
// the MFMAs of tile t and written to LDS after them -- into the other buffer with one barrier per key tile (NBUF = 2, head dims
// <= 64), or into the same buffer between two barriers (NBUF = 1: the wide heads, whose tiles would not leave room for two
// workgroups per CU).  With QT = 2 the K / V^T fragments read from LDS feed two query tiles (half the LDS traffic per MFMA).
// Scheduling hint for a region that holds NM MFMAs and independent VALU work: NM groups of [1 MFMA, NV VALU instructions]
template <int G, int NM, int NV>
__device__ __forceinline__ void mfma_valu_pattern() {
  if constexpr (G < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    mfma_valu_pattern<G + 1, NM, NV>();
  }
}

template <int DQ, int DV, int NS, int QT, int NBUF>
__global__ __launch_bounds__(256, 1) void attn_kernel(const u16* __restrict__ q_hi, const u16* __restrict__ q_lo,
                                                   const u16* __restrict__ k_hi, const u16* __restrict__ k_lo,
                                                   const u16* __restrict__ vt_hi, const u16* __restrict__ vt_lo,
                                                   u16* __restrict__ out_sp, int ldo, int H, int L, int Lk, int Lpad, int dhead) {
  constexpr int NPL = NS >= 3 ? 2 : 1;
  constexpr int KP = DQ + 8;       // LDS pitch of a K row (16-bit elements)
  constexpr int VP = KV_TILE + 8;  // LDS pitch of a V^T row
  constexpr int QS = DQ / 32;      // full MFMA k-steps (32 channels) over the head dim
  constexpr bool TAIL = (DQ % 32) != 0;   // + one 16-deep step: head dims 40 -> 48 = 32 + 16, 80 = 64 + 16, 4 -> 16 (no padding to 32 / 64)
  static_assert(DQ % 16 == 0, "q / k rows are padded to a multiple of 16 channels");
  constexpr int DT = DV / 16;      // output d-tiles
  constexpr int KCH = KV_TILE * DQ / 8;            // 16-byte chunks of a K tile (per plane)
  constexpr int VCH = DV * 8;                      // 16-byte chunks of a V^T tile (per plane)
  constexpr int KREG = (KCH + 255) / 256, VREG = (VCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) u16 sK[NBUF][NPL][KV_TILE][KP];
  __shared__ __attribute__((aligned(16))) u16 sV[NBUF][NPL][DV][VP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  // XCD-aware block order: workgroup i runs on XCD i % 8 (private L2 each).  Hand every XCD a contiguous range of
  // (batch, head, query block) so that the query blocks of one head -- which all stream the same K / V^T planes --
  // share one L2 instead of fetching them into all eight (measured: 262 MB fetched per launch for 57 MB of operands).
  int logical;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qblocks = (Lpad + 64 * QT - 1) / (64 * QT);
  const int bhi = logical / qblocks;
  const int b = bhi / H, h = bhi - b * H;
  const int q0 = (logical - bhi * qblocks) * 64 * QT + wave * 16 * QT;      // first query of this wave
  const size_t bh = (size_t)b * H + h;
  const u16* kp[2] = {k_hi + bh * Lpad * DQ, k_lo + bh * Lpad * DQ};
  const u16* vp[2] = {vt_hi + bh * DV * Lpad, vt_lo + bh * DV * Lpad};

  bool active[QT];
  op16x8 qh[QT][QS > 0 ? QS : 1], ql[QT][QS > 0 ? QS : 1];
  op4_t qth[QT], qtl[QT];              // the 16-channel tail of the query rows
#pragma unroll
  for (int t2 = 0; t2 < QT; ++t2) {
    active[t2] = q0 + 16 * t2 < L;
    const bool inb = q0 + 16 * t2 + c < Lpad;           // rows past Lpad belong to the next head's planes: never read them
    const size_t qoff = (bh * Lpad + q0 + 16 * t2 + c) * DQ + g * 8;
#pragma unroll
    for (int ks = 0; ks < QS; ++ks) {
      qh[t2][ks] = inb ? *(const op16x8*)(q_hi + qoff + ks * 32) : (op16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (NS >= 3) ql[t2][ks] = inb ? *(const op16x8*)(q_lo + qoff + ks * 32) : (op16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (TAIL) {      // channels 32 QS + 4g .. + 3 (qoff carries 8g: the tail slice starts at 4g)
      const size_t toff = (bh * Lpad + q0 + 16 * t2 + c) * DQ + QS * 32 + g * 4;
      qth[t2] = inb ? *(const op4_t*)(q_hi + toff) : (op4_t){0, 0, 0, 0};
      if (NS >= 3) qtl[t2] = inb ? *(const op4_t*)(q_lo + toff) : (op4_t){0, 0, 0, 0};
    }
  }
  f32x4 o[QT][DT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t2 = 0; t2 < QT; ++t2) {
#pragma unroll
    for (int i = 0; i < DT; ++i) o[t2][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m_run[t2] = -INFINITY;
    l_run[t2] = 0.f;
  }

  // ---- staging: each thread owns fixed 16-byte chunks of the K tile / V^T tile (both planes)
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;      // (HIP's uint4 is a struct-of-union: arrays of it stay in memory)
  u32x4 kr[NPL][KREG], vr[NPL][VREG];
  auto load_tile = [&](int kv0) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < KREG; ++i) {       // (threads past the tile re-load its last chunk: every register is always defined)
        const int ch = min(tid + i * 256, KCH - 1);
        const int row = ch / (DQ / 8), kc = ch - row * (DQ / 8);
        kr[pl][i] = *(const u32x4*)(kp[pl] + ((size_t)(kv0 + row)) * DQ + kc * 8);
      }
#pragma unroll
      for (int i = 0; i < VREG; ++i) {
        const int ch = min(tid + i * 256, VCH - 1);
        const int row = ch >> 3, kc = ch & 7;
        vr[pl][i] = *(const u32x4*)(vp[pl] + (size_t)row * Lpad + kv0 + kc * 8);
      }
    }
  };
  auto write_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < KREG; ++i) {       // (threads past the tile store the last chunk again: same address, same value)
        const int ch = min(tid + i * 256, KCH - 1);
        const int row = ch / (DQ / 8), kc = ch - row * (DQ / 8);
        *(u32x4*)&sK[buf][pl][row][kc * 8] = kr[pl][i];
      }
#pragma unroll
      for (int i = 0; i < VREG; ++i) {
        const int ch = min(tid + i * 256, VCH - 1);
        const int row = ch >> 3, kc = ch & 7;
        *(u32x4*)&sV[buf][pl][row][kc * 8] = vr[pl][i];
      }
    }
  };

  const int ntiles = (Lk + KV_TILE - 1) / KV_TILE;     // keys [0, Lk) take part (Lk <= L: trailing padding tokens are masked)
  load_tile(0);
  write_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * KV_TILE, buf = NBUF == 2 ? (t & 1) : 0;
    if (t + 1 < ntiles) load_tile(kv0 + KV_TILE);      // in flight under the MFMAs below

    // (Round 4 built a PHASED form of the two-query-tile loop -- S(q0) | S(q1) with softmax(q0) | PV(q0) with softmax(q1) | PV(q1) -- and
    //  measured it SLOWER: attn_kernel<48,48,3,2,2> 72.4 us vs 64.3 us per launch at L = 1024, 19.0 vs 16.4 ms of attention per step at
    //  L = 4096 (profiles/r04_step_trace_v4.txt): two co-resident workgroups already overlap one's softmax with the other's MFMAs.  It
    //  lived behind -DMVD_ATTN_PHASED until round 6 and is in the history at f2958d7.)
    // ---- S^T = K Q^T for every query tile of the wave: s[t2][kt][r] = S[q = c of tile t2][key = kv0 + kt*16 + g*4 + r].  The K
    //      fragments are read from LDS ONCE per key sub-tile and feed all QT query tiles (tiles past L compute on zero / padding rows
    //      and are never stored: no branches in the loop body, one basic block to schedule).
    f32x4 s[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int t2 = 0; t2 < QT; ++t2) s[t2][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < QS; ++ks) {
        const op16x8 kh = *(const op16x8*)&sK[buf][0][kt * 16 + c][ks * 32 + g * 8];
        if (NS >= 3) {
          const op16x8 kl = *(const op16x8*)&sK[buf][NPL - 1][kt * 16 + c][ks * 32 + g * 8];
          if (NS == 4) {
#pragma unroll
            for (int t2 = 0; t2 < QT; ++t2) s[t2][kt] = MVD_MFMA_16x16x32(kl, ql[t2][ks], s[t2][kt], 0, 0, 0);
          }
#pragma unroll
          for (int t2 = 0; t2 < QT; ++t2) s[t2][kt] = MVD_MFMA_16x16x32(kl, qh[t2][ks], s[t2][kt], 0, 0, 0);
#pragma unroll
          for (int t2 = 0; t2 < QT; ++t2) s[t2][kt] = MVD_MFMA_16x16x32(kh, ql[t2][ks], s[t2][kt], 0, 0, 0);
        }
#pragma unroll
        for (int t2 = 0; t2 < QT; ++t2) s[t2][kt] = MVD_MFMA_16x16x32(kh, qh[t2][ks], s[t2][kt], 0, 0, 0);
      }
    }
    // the 16-channel tail as its own pass over the four key sub-tiles: consecutive MFMAs hit different accumulators.  (Issued right
    // behind the last 16x16x32 step of the SAME accumulator, the 16x16x16 MFMA read a stale value in the one-product (NS = 1)
    // 80-channel instantiation -- tests/test_gpu_ops.py::test_qkv_gemm_and_attention[2-8-256-80-1].)
    if (TAIL) {
      op4_t kh4[4], kl4[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        kh4[kt] = *(const op4_t*)&sK[buf][0][kt * 16 + c][QS * 32 + g * 4];
        if (NS >= 3) kl4[kt] = *(const op4_t*)&sK[buf][NPL - 1][kt * 16 + c][QS * 32 + g * 4];
      }
#pragma unroll
      for (int t2 = 0; t2 < QT; ++t2) {
        if (NS == 4) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) s[t2][kt] = MVD_MFMA_16x16x16(kl4[kt], qtl[t2], s[t2][kt]);
        }
        if (NS >= 3) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) s[t2][kt] = MVD_MFMA_16x16x16(kl4[kt], qth[t2], s[t2][kt]);
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) s[t2][kt] = MVD_MFMA_16x16x16(kh4[kt], qtl[t2], s[t2][kt]);
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[t2][kt] = MVD_MFMA_16x16x16(kh4[kt], qth[t2], s[t2][kt]);
      }
    }
    // ---- online softmax (per query column c; keys are spread over registers and the 4 lane groups)
    // (q carries dhead^-0.5 * log2(e) from the QKV epilogue, so the scores are base-2 logits: exp2 below is a bare v_exp_f32)
    op16x8 ph[QT][2], pl2[QT][2];
#pragma unroll
    for (int t2 = 0; t2 < QT; ++t2) {
      if (kv0 + KV_TILE > Lk) {     // ragged last tile only (uniform branch): keys past Lk do not take part
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + kt * 16 + g * 4 + r >= Lk) s[t2][kt][r] = -INFINITY;
      }
      float mt = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[t2][kt][r]);
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run[t2], mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run[t2] - m_new);
      m_run[t2] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[t2][kt][r] = __builtin_amdgcn_exp2f(s[t2][kt][r] - m_new);
          psum += s[t2][kt][r];
        }
      l_run[t2] = l_run[t2] * alpha + psum;
#pragma unroll
      for (int i = 0; i < DT; ++i) o[t2][i] *= alpha;
      // ---- P^T as MFMA B operand: k-slot (g, j): j<4 -> key 32u + 4g + j ; j>=4 -> key 32u + 16 + 4g + (j-4)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        union { op16x8 v; u16 e[8]; uint32_t w[4]; } H8, L8;
        if (NS >= 3) {     // packed split: v_cvt_pk + v_pk_add, 5 instructions per pair (common.hpp: split_op16x2)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4& sv = j < 2 ? s[t2][2 * u] : s[t2][2 * u + 1];
            split_op16x2(sv[2 * (j & 1)], sv[2 * (j & 1) + 1], H8.w[j], L8.w[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) H8.e[j] = to_op_bits(j < 4 ? s[t2][2 * u][j] : s[t2][2 * u + 1][j - 4]);
        }
        ph[t2][u] = H8.v;
        if (NS >= 3) pl2[t2][u] = L8.v;
      }
    }
    // ---- O^T += V^T P^T: every V^T fragment is read once and multiplies the P^T of all QT query tiles
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        union { op16x8 v; uint2 h2[2]; } VH, VL;
        VH.h2[0] = *(const uint2*)&sV[buf][0][dt * 16 + c][32 * u + 4 * g];
        VH.h2[1] = *(const uint2*)&sV[buf][0][dt * 16 + c][32 * u + 16 + 4 * g];
        if (NS >= 3) {
          VL.h2[0] = *(const uint2*)&sV[buf][NPL - 1][dt * 16 + c][32 * u + 4 * g];
          VL.h2[1] = *(const uint2*)&sV[buf][NPL - 1][dt * 16 + c][32 * u + 16 + 4 * g];
          if (NS == 4) {
#pragma unroll
            for (int t2 = 0; t2 < QT; ++t2) o[t2][dt] = MVD_MFMA_16x16x32(VL.v, pl2[t2][u], o[t2][dt], 0, 0, 0);
          }
#pragma unroll
          for (int t2 = 0; t2 < QT; ++t2) o[t2][dt] = MVD_MFMA_16x16x32(VL.v, ph[t2][u], o[t2][dt], 0, 0, 0);
#pragma unroll
          for (int t2 = 0; t2 < QT; ++t2) o[t2][dt] = MVD_MFMA_16x16x32(VH.v, pl2[t2][u], o[t2][dt], 0, 0, 0);
        }
#pragma unroll
        for (int t2 = 0; t2 < QT; ++t2) o[t2][dt] = MVD_MFMA_16x16x32(VH.v, ph[t2][u], o[t2][dt], 0, 0, 0);
      }
    }
    if (NBUF == 2) {
      if (t + 1 < ntiles) write_tile(buf ^ 1);         // the other buffer: its readers finished before the previous barrier
      __syncthreads();
    } else {
      __syncthreads();                                 // everyone is done reading the single buffer
      if (t + 1 < ntiles) write_tile(0);
      __syncthreads();
    }
  }
#pragma unroll
  for (int t2 = 0; t2 < QT; ++t2) {
    if (!active[t2]) continue;
    float l_tot = l_run[t2] + __shfl_xor(l_run[t2], 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + 16 * t2 + c;
    if (q < L) {
      const size_t orow = (size_t)b * L + q;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d0 = dt * 16 + g * 4;
        if (d0 < dhead)
          store_sp4(out_sp, orow, ldo, h * dhead + d0, o[t2][dt][0] * inv, o[t2][dt][1] * inv, o[t2][dt][2] * inv, o[t2][dt][3] * inv);
      }
    }
  }
}

// one wave per pixel; lanes stride over channels of a head. q (P,C), k/v (P*D, C), D <= 8.
__global__ __launch_bounds__(256) void pixel_xattn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, u16* __restrict__ out_sp, int P, int D,
                                                          int heads, int dhead) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= P) return;
  const int C = heads * dhead;
  const float scale = rsqrtf((float)dhead);
  for (int h = 0; h < heads; ++h) {
    float sc[8];
    for (int j = 0; j < D; ++j) {
      float part = 0.f;
      for (int e = lane; e < dhead; e += 64)
        part += q[(size_t)pix * C + h * dhead + e] * k[((size_t)pix * D + j) * C + h * dhead + e];
      sc[j] = wave_sum(part) * scale;
    }
    float mx = sc[0];
    for (int j = 1; j < D; ++j) mx = fmaxf(mx, sc[j]);
    float den = 0.f;
    for (int j = 0; j < D; ++j) {
      sc[j] = expf(sc[j] - mx);
      den += sc[j];
    }
    for (int e = lane; e < dhead; e += 64) {
      float acc = 0.f;
      for (int j = 0; j < D; ++j) acc += (sc[j] / den) * v[((size_t)pix * D + j) * C + h * dhead + e];
      store_sp1(out_sp, (size_t)pix, C, h * dhead + e, acc);
    }
  }
}

// timm Attention core over the V reference views: one wavefront per sequence.  The V x 3C fp32 rows of the sequence
// are read once with coalesced 16-byte loads into a wave-private LDS slab; lane (h, vq) then owns one query of one
// head: V dot products of dhead = 32, softmax over the V keys, weighted sum of the V value rows, split-plane store.
// qkv row layout [3][heads][dhead] (timm reshape B,N,3,H,hd).  Requires heads * V <= 64 and dhead == 32.
template <int V>
__global__ __launch_bounds__(256) void view_mha_kernel(const float* __restrict__ qkv, u16* __restrict__ out_sp, int Nseq, int heads,
                                                       int dhead) {
  constexpr int C = 256, ROW = 3 * C;
  __shared__ __attribute__((aligned(16))) float s[4][V * ROW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t n = (size_t)blockIdx.x * 4 + w;
  if (n >= (size_t)Nseq) return;
  const float4* src = (const float4*)(qkv + n * V * ROW);
  float4* dst = (float4*)s[w];
#pragma unroll
  for (int i = 0; i < V * ROW / 4 / 64; ++i) dst[lane + 64 * i] = src[lane + 64 * i];
  // (wave-private slab: no barrier needed, LDS operations of one wave complete in order)
  const float scale = rsqrtf((float)dhead);
  for (int item = lane; item < heads * V; item += 64) {
    const int h = item / V, vq = item - h * V;
    const float* q = s[w] + vq * ROW + h * 32;
    float sc[V];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float* k = s[w] + j * ROW + C + h * 32;
      float a = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < 8; ++d4) {
        const float4 qq = *(const float4*)(q + d4 * 4), kk = *(const float4*)(k + d4 * 4);
        a += (qq.x * scale) * kk.x;
        a += (qq.y * scale) * kk.y;
        a += (qq.z * scale) * kk.z;
        a += (qq.w * scale) * kk.w;
      }
      sc[j] = a;
      mx = fmaxf(mx, a);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      sc[j] = expf(sc[j] - mx);
      den += sc[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) sc[j] = sc[j] / den;
    const size_t orow = n * V + vq;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float4 vv = *(const float4*)(s[w] + j * ROW + 2 * C + h * 32 + d4 * 4);
        o.x += sc[j] * vv.x;
        o.y += sc[j] * vv.y;
        o.z += sc[j] * vv.z;
        o.w += sc[j] * vv.w;
      }
      store_sp4(out_sp, orow, C, h * 32 + d4 * 4, o.x, o.y, o.z, o.w);
    }
  }
}

// generic fallback (any V <= 16, any head size): one thread per (sequence, head, query view)
__global__ __launch_bounds__(256) void view_mha_generic_kernel(const float* __restrict__ qkv, u16* __restrict__ out_sp, int Nseq, int V,
                                                               int heads, int dhead) {
  const size_t total = (size_t)Nseq * heads * V;
  const int C = heads * dhead;
  const float scale = rsqrtf((float)dhead);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int vq = (int)(e % V);
    const int h = (int)((e / V) % heads);
    const size_t n = e / ((size_t)V * heads);
    const float* qrow = qkv + (n * V + vq) * 3 * C + h * dhead;
    float sc[16];
    float mx = -INFINITY;
    for (int j = 0; j < V; ++j) {
      const float* krow = qkv + (n * V + j) * 3 * C + C + h * dhead;
      float a = 0.f;
      for (int d = 0; d < dhead; ++d) a += (qrow[d] * scale) * krow[d];
      sc[j] = a;
      mx = fmaxf(mx, a);
    }
    float den = 0.f;
    for (int j = 0; j < V; ++j) {
      sc[j] = expf(sc[j] - mx);
      den += sc[j];
    }
    const size_t orow = n * V + vq;
    for (int d = 0; d < dhead; ++d) {
      float a = 0.f;
      for (int j = 0; j < V; ++j) a += (sc[j] / den) * qkv[(n * V + j) * 3 * C + 2 * C + h * dhead + d];
      store_sp1(out_sp, orow, C, h * dhead + d, a);
    }
  }
}

// one wave per sequence: logits w.x_v + b, softmax over V, out = sum_v p_v x_v.  C % 64 == 0, C <= 512.
__global__ __launch_bounds__(256) void view_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, u16* __restrict__ out_sp, int Nseq, int V,
                                                        int C) {
  const int lane = threadIdx.x & 63;
  const size_t n = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= (size_t)Nseq) return;
  float lg[16];
  float mx = -INFINITY;
  for (int v = 0; v < V; ++v) {
    float part = 0.f;
    for (int e = lane; e < C; e += 64) part += x[(n * V + v) * C + e] * w[e];
    lg[v] = wave_sum(part) + bias[0];
    mx = fmaxf(mx, lg[v]);
  }
  float den = 0.f;
  for (int v = 0; v < V; ++v) {
    lg[v] = expf(lg[v] - mx);
    den += lg[v];
  }
  for (int e = lane; e < C; e += 64) {
    float a = 0.f;
    for (int v = 0; v < V; ++v) a += x[(n * V + v) * C + e] * (lg[v] / den);
    store_sp1(out_sp, n, C, e, a);
  }
}

template <int NS>
int launch_attn(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi, const void* vt_lo,
                void* out_sp, int ldo, int B, int H, int L, int Lk, int Lpad, int dhead, hipStream_t s) {
  const int dq = mvd_attn_dpad(dhead), dv = mvd_attn_dpad(dhead);
  // two query tiles per wavefront once that still leaves >= 2 workgroups per CU (long sequences); small heads only
  const long wg2 = (long)((Lpad + 127) / 128) * H * B;
  // (MVD_ATTN_QT1=1: debugging knob -- always the one-tile-per-wavefront kernel, whose per-query arithmetic the two-tile kernel must
  //  reproduce bit for bit: tests/test_gpu_ops.py::test_attention_phased_equals_single_tile)
  const bool two = dq <= 96 && wg2 >= 512 && getenv("MVD_ATTN_QT1") == nullptr;
  const dim3 block(256);
#define MVD_ATTN_CASE(DQ, DV)                                                                                            \
  if (dq == DQ && dv == DV) {                                                                                            \
    if (two && DQ <= 96)                                                                                                  \
      hipLaunchKernelGGL((attn_kernel<DQ, DV, NS, (DQ <= 96 ? 2 : 1), (DQ <= 64 ? 2 : 1)>), dim3((unsigned)wg2), block, 0, s, (const u16*)q_hi, \
                         (const u16*)q_lo, (const u16*)k_hi, (const u16*)k_lo, (const u16*)vt_hi, (const u16*)vt_lo, (u16*)out_sp,  \
                         ldo, H, L, Lk, Lpad, dhead);                                                                    \
    else                                                                                                                 \
      hipLaunchKernelGGL((attn_kernel<DQ, DV, NS, 1, (DQ <= 64 ? 2 : 1)>), dim3((Lpad / 64) * H * B), block, 0, s, (const u16*)q_hi, \
                         (const u16*)q_lo, (const u16*)k_hi, (const u16*)k_lo, (const u16*)vt_hi, (const u16*)vt_lo, (u16*)out_sp,  \
                         ldo, H, L, Lk, Lpad, dhead);                                                                    \
    return 0;                                                                                                            \
  }
  MVD_ATTN_CASE(16, 16)
  MVD_ATTN_CASE(32, 32)
  MVD_ATTN_CASE(48, 48)
  MVD_ATTN_CASE(64, 64)
  MVD_ATTN_CASE(80, 80)
  MVD_ATTN_CASE(160, 160)
#undef MVD_ATTN_CASE
  return -1;
}

}  // namespace

extern "C" int mvd_attn_lpad(int L) { return (L + 63) & ~63; }
extern "C" size_t mvd_attn_qk_plane_elems(int B, int heads, int L, int dhead) {
  return (size_t)B * heads * mvd_attn_lpad(L) * mvd_attn_dpad(dhead);
}
extern "C" size_t mvd_attn_vt_plane_elems(int B, int heads, int L, int dhead) {
  return (size_t)B * heads * mvd_attn_lpad(L) * mvd_attn_dpad(dhead);
}

extern "C" int mvd_attention(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                             const void* vt_lo, void* out_sp, int ldo, int B, int heads, int L, int Lkeys, int dhead, int prec,
                             mvd_stream_t stream) {
  MVD_CHECK_ARG(q_hi && q_lo && k_hi && k_lo && vt_hi && vt_lo && out_sp, "mvd_attention: null pointer");
  MVD_CHECK_ARG(B > 0 && heads > 0 && L > 0 && dhead > 0 && dhead % 4 == 0, "mvd_attention: bad shape (dhead %% 4 == 0)");
  MVD_CHECK_ARG(ldo % 32 == 0 && ((uintptr_t)out_sp & 127) == 0, "mvd_attention: out must be split planes (ldo %% 32 == 0, 128-byte aligned)");
  MVD_CHECK_ARG(prec == MVD_PREC_X1 || prec == MVD_PREC_X3 || prec == MVD_PREC_X4, "mvd_attention: bad prec");
  const int Lpad = mvd_attn_lpad(L);
  const int Lk = Lkeys > 0 ? Lkeys : L;
  MVD_CHECK_ARG(Lk <= L, "mvd_attention: Lkeys=%d must be <= L=%d", Lk, L);
  int rc;
  if (prec == MVD_PREC_X4)
    rc = launch_attn<4>(q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, out_sp, ldo, B, heads, L, Lk, Lpad, dhead, (hipStream_t)stream);
  else if (prec == MVD_PREC_X3)
    rc = launch_attn<3>(q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, out_sp, ldo, B, heads, L, Lk, Lpad, dhead, (hipStream_t)stream);
  else
    rc = launch_attn<1>(q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, out_sp, ldo, B, heads, L, Lk, Lpad, dhead, (hipStream_t)stream);
  MVD_CHECK_ARG(rc == 0, "mvd_attention: unsupported head dim %d (supported: <=32, 33..64, 65..96 with dv 80, 160)", dhead);
  MVD_CHECK_LAUNCH("mvd_attention");
  return 0;
}

extern "C" int mvd_pixel_cross_attn(const float* q, const float* k, const float* v, void* out_sp, int P, int D, int heads,
                                    int dhead, mvd_stream_t stream) {
  MVD_CHECK_ARG(q && k && v && out_sp && (heads * dhead) % 32 == 0 && P > 0 && D > 0 && D <= 8 && heads > 0 && dhead > 0, "mvd_pixel_cross_attn: bad arguments (D <= 8)");
  hipLaunchKernelGGL(pixel_xattn_kernel, dim3(cdiv(P, 4)), dim3(256), 0, (hipStream_t)stream, q, k, v, (u16*)out_sp, P, D, heads, dhead);
  MVD_CHECK_LAUNCH("mvd_pixel_cross_attn");
  return 0;
}

extern "C" int mvd_view_mha(const float* qkv, void* out_sp, int Nseq, int V, int heads, int dhead, mvd_stream_t stream) {
  MVD_CHECK_ARG(qkv && out_sp && (heads * dhead) % 32 == 0 && Nseq > 0 && V > 0 && V <= 16 && heads > 0 && dhead > 0,
                "mvd_view_mha: bad arguments (V <= 16)");
  hipStream_t s = (hipStream_t)stream;
  const bool fast = heads == 8 && dhead == 32 && ((uintptr_t)qkv & 15) == 0;
  const dim3 grid(cdiv(Nseq, 4)), block(256);
#define MVD_VMHA(VV)                                                                                     \
  if (fast && V == VV) {                                                                                 \
    hipLaunchKernelGGL(view_mha_kernel<VV>, grid, block, 0, s, qkv, (u16*)out_sp, Nseq, heads, dhead);   \
    MVD_CHECK_LAUNCH("mvd_view_mha");                                                                    \
    return 0;                                                                                            \
  }
  MVD_VMHA(2) MVD_VMHA(3) MVD_VMHA(4) MVD_VMHA(8)
#undef MVD_VMHA
  const size_t total = (size_t)Nseq * heads * V;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(view_mha_generic_kernel, dim3(blocks), block, 0, s, qkv, (u16*)out_sp, Nseq, V, heads, dhead);
  MVD_CHECK_LAUNCH("mvd_view_mha");
  return 0;
}

extern "C" int mvd_view_pool(const float* x, const float* w, const float* b, void* out_sp, int Nseq, int V, int C,
                             mvd_stream_t stream) {
  MVD_CHECK_ARG(x && w && b && out_sp && C % 32 == 0 && Nseq > 0 && V > 0 && V <= 16 && C > 0, "mvd_view_pool: bad arguments (V <= 16)");
  hipLaunchKernelGGL(view_pool_kernel, dim3(cdiv(Nseq, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, (u16*)out_sp, Nseq, V, C);
  MVD_CHECK_LAUNCH("mvd_view_pool");
  return 0;
}
