// mvd_ffchain: the feed-forward chain of a transformer block at C = 320 as ONE tile kernel (+ the split reduce of its output) --
//     out = [ GEGLU(LN(t2) W1^T + b1) | t2 ] Wm^T (+ bias, residual, statistics: the epilogue of the second GEMM's descriptor)
// i.e. FeedForward (attention.py:37-64) folded with norm3 (attention.py:223) in front and proj_out behind it (attention.py:259 /
// mvdfusion/attention.py:114; the merged weight Wm = [Wp W2 | Wp] of attention.py::compose_ff_out_proj).  Today these are two mvd_gemm
// launches with the (M, 4C) GEGLU output going through memory between them.
//
// A workgroup (8 waves) owns a 64-row panel of t2 and half of the 1280 inner columns:
//   * the panel's operand planes (64 x 320 hi + lo = 80 KiB) are DMA'd into LDS once and stay there;
//   * per chunk of 64 inner columns: ten k-tiles of  H^T (128 packed value | gate columns x 64 rows) += W1c x^T  -- the WEIGHT fragment is the
//     MFMA's A operand and the activation fragment its B operand, so a lane ends up with 4 consecutive features of one row -- then the
//     LayerNorm fold, bias and GEGLU in registers (value and gate of a feature sit in the same lane), h split into hi + lo and written
//     as 8-byte pieces into an A-layout LDS tile (64 rows x 64 k), then four stages of  out^T (160 columns x 64 rows) += W2c h^T;
//   * the last five k-tiles of the merged weight multiply the resident panel itself (the "| t2" part), shared between the two halves;
//   * weights stream through a ring of three 20 KiB LDS slots with counted vmcnt waits (two stages in flight), the per-chunk fold
//     vectors (column sums and biases of the chunk's value / gate features) arrive by one 1 KiB LDS-DMA of wave 0;
//   * the 64 x 320 partial output goes to a split-K slab; the descriptor's own split reduce applies its epilogue (mvd_gemm's reduce
//     kernels, the fused reduce + GroupNorm among them).
// Same products as the two-GEMM path (x_lo w_hi + x_hi w_lo + x_hi w_hi per k-step), another summation order over the inner dimension.
#include "gemm_device.hpp"
#ifdef FF_PROBE_NOMFMA      // (timing probe: the loop without its MFMAs -- wrong results)
#define FF_MFMA(a, b, c) (c)
#else
#define FF_MFMA(a, b, c) MVD_MFMA_16x16x32(a, b, c, 0, 0, 0)
#endif

namespace {

constexpr int FF_BM = 64, FF_C = 320, FF_KT = FF_C / 32, FF_NST = 150;
#ifndef FF_PF
#define FF_PF 6        // stages the L2 prefetcher (wave 8) runs ahead; 0 = no prefetcher wave
#endif
constexpr int FF_THREADS = FF_PF > 0 ? 576 : 512;
constexpr int FF_XP = FF_KT * 8192, FF_HT = 2 * 8192, FF_SLOT = 20480, FF_VEC = 2048;
constexpr int FF_SMEM = FF_XP + FF_HT + 3 * FF_SLOT + FF_VEC + 512;
static_assert(FF_SMEM <= 160 * 1024, "LDS budget");

struct FfParams {
  mvd_gemm_desc g1, g2;
  float* slabs;
};

template <int N>
__device__ __forceinline__ void ff_wait_barrier() {
#ifdef FF_PROBE_NOWAIT      // (timing probe: never wait for the DMAs -- wrong results, shows the loop without the delivery latency)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory");
#endif
}

__device__ __forceinline__ void ff_dma16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__global__ __launch_bounds__(FF_THREADS, 1) void ffchain_kernel(FfParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[FF_SMEM];
  unsigned char* const xP = smem;
  unsigned char* const hT = smem + FF_XP;
  unsigned char* const ring = hT + FF_HT;
  float* const vecs = (float*)(ring + 3 * FF_SLOT);
  float* const s_rows = (float*)(ring + 3 * FF_SLOT + FF_VEC);
  const mvd_gemm_desc& d1 = p.g1;
  const mvd_gemm_desc& d2 = p.g2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * FF_BM, split = blockIdx.y;
  const int M = d1.M;
  const int nt1 = d1.N >> 4, nt2 = d2.N >> 4, half1 = d1.N >> 1;
  const u16* zero = (const u16*)g_zero_page;
  const int gq = lane >> 4, c = lane & 15;

  // ---- wave 8: the L2 prefetcher.  All 256 workgroups sweep the same 2.4 MB of weights in step, so the LDS-DMA of a stage is the first touch of its
  //      lines in this XCD's L2.  This wave only takes part in the barriers and requests every 128-byte line of stage t + FF_PF with plain loads
  //      whose result nobody waits for (one dword per line into ONE register that stays reserved to the end).
  if (FF_PF > 0 && wave == 8) {
    unsigned sink = 0;
    auto prefetch = [&](int t) {
      if (t >= FF_NST) return;
      const char* base;
      int lines;
      if (t < 140 && (t % 14) < 10) {
        base = (const char*)((const u16*)d1.Wp + ((size_t)(t % 14) * nt1 + (split * 10 + t / 14) * 8) * 1024);
        lines = 128;
      } else {
        int k2, hf;
        if (t < 140) {
          const int q = t % 14 - 10;
          k2 = (split * 10 + t / 14) * 2 + (q >> 1);
          hf = q & 1;
        } else {
          k2 = 40 + split * 5 + ((t - 140) >> 1);
          hf = (t - 140) & 1;
        }
        base = (const char*)((const u16*)d2.Wp + ((size_t)k2 * nt2 + hf * 10) * 1024);
        lines = 160;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int ln = min(lane + 64 * i, lines - 1);
        const char* a = base + ln * 128;
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(a) : "memory");      // ("+v": one destination register for every request)
      }
    };
    for (int t = 0; t < FF_PF; ++t) prefetch(t);
#pragma unroll 1
    for (int t = 0; t < FF_NST; ++t) {
      asm volatile("s_barrier" ::: "memory");
      prefetch(t + FF_PF);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(sink) : "memory");
    return;
  }

  // ---- the panel: 80 granules (8 rows x 128 B), wave w takes the 8-row group w of every k-tile
  {
    const int gr = lane >> 3, R = (wave & 1) * 8 + gr, gc = (lane & 7) ^ ((R >> 1) & 7);
    const int m = m0 + wave * 8 + gr;
    const u16* src = m < M ? (const u16*)d1.A + (size_t)m * 2 * d1.lda + gc * 8 : zero;
    const int step = m < M ? 64 : 0;
#pragma unroll
    for (int kt = 0; kt < FF_KT; ++kt) ff_dma16(src + kt * step, xP + kt * 8192 + wave * 1024);
  }
  if (tid < FF_BM) {
    const float2 st = m0 + tid < M ? ln_row_stats(d1, m0 + tid) : make_float2(0.f, 0.f);
    s_rows[tid * 2] = st.x;
    s_rows[tid * 2 + 1] = st.y;
  }

  // ---- stage sources.  A stage is a run of whole 1 KiB fragment images, contiguous in the packed weight: granule gi of a stage is at
  //      (stage base) + 1024 gi in memory AND in its LDS slot, so a wave's DMA i moves bytes [voff + 8192 i, + 1024) of the stage.
  //      W1 stage (chunk cc, k-tile kt): 16 granules; W2 stage (chunk cc, q): 20 granules, k-tile 2 cc + q / 2, output half q & 1;
  //      the "| t2" part (q = 0 .. 9): W2 k-tiles 40 + 5 split + q / 2.  Everything but `voff` is uniform (scalar registers).
  const unsigned voff = wave * 1024 + lane * 16;
  const size_t k1s = (size_t)nt1 * 2048, k2s = (size_t)nt2 * 2048;
  const char* const w1b = (const char*)d1.Wp + (size_t)(split * 10) * 16384;
  const char* const w2b = (const char*)d2.Wp + (size_t)(split * 10) * 2 * k2s;
  const char* const w2x = (const char*)d2.Wp + (size_t)(40 + split * 5) * k2s;
  const int n2 = wave < 4 ? 3 : 2;               // DMAs of this wave for a 20-granule stage
  auto dma_stage = [&](const char* sbase, unsigned char* slot, int n) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < n) ff_dma16(sbase + voff + 8192 * i, slot + (wave + 8 * i) * 1024);
  };
  auto issue_vecs = [&](int cidx) {              // wave 0: column sums / biases of chunk cidx's 64 value and 64 gate features -> vecs[parity]
    const int f0 = (split * 10 + cidx) * 64 + c * 4;
    const float* src = gq == 0 ? d1.ln_colsum + f0 : (gq == 1 ? d1.ln_colsum + half1 + f0 : (gq == 2 ? d1.bias + f0 : d1.bias + half1 + f0));
    if (gq >= 2 && d1.bias == nullptr) src = (const float*)g_zero_page;
    ff_dma16(src, (unsigned char*)vecs + (cidx & 1) * 1024);
  };

  // fragment read offsets of an activation tile in the A layout (gemm_plain.hpp): row = lane & 15 of a 16-row block, hi chunk = lane >> 4
  const int fsw = (c >> 1) & 7;
  const int fbase = (c >> 3) * 1024 + (c & 7) * 128;
  const int foff_hi = fbase + ((gq ^ fsw) << 4);
  const int foff_lo = fbase + (((4 + gq) ^ fsw) << 4);

  const int wm = wave >> 2, wn = wave & 3;        // phase 1: 2 x 4 waves, wave tile 32 rows x 32 packed columns (one value | gate block)
  const int wm4 = wave >> 1, wn2 = wave & 1;      // phase 2: 4 x 2 waves, wave tile 16 rows x 80 columns of the current output half
  f32x4 acc1[2][2], acc2[2][5];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc2[h][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float scale1 = gemm_acc_scale(d1);

  auto phase2 = [&](auto half_c, const unsigned char* slot, const unsigned char* act) {
    constexpr int HF = decltype(half_c)::value;
    const op16x8 bh = *(const op16x8*)(act + foff_hi), bl = *(const op16x8*)(act + foff_lo);
    op16x8 wh[5], wl[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      wh[j] = *(const op16x8*)(slot + ((wn2 * 5 + j) * 2) * 1024 + lane * 16);
      wl[j] = *(const op16x8*)(slot + ((wn2 * 5 + j) * 2 + 1) * 1024 + lane * 16);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) acc2[HF][j] = FF_MFMA(wh[j], bl, acc2[HF][j]);
#pragma unroll
    for (int j = 0; j < 5; ++j) acc2[HF][j] = FF_MFMA(wl[j], bh, acc2[HF][j]);
#pragma unroll
    for (int j = 0; j < 5; ++j) acc2[HF][j] = FF_MFMA(wh[j], bh, acc2[HF][j]);
  };

  using std::integral_constant;
  unsigned char *sl0 = ring, *sl1 = ring + FF_SLOT, *sl2 = ring + 2 * FF_SLOT;      // slot of the stage computed now / next / staged now
  auto wait_next = [&](bool next_w1, bool extra) {
    // this wave's DMAs of the current stage have landed once only the NEXT stage's (2, or 3 on waves 0-3 for a 20-granule stage) are
    // outstanding -- plus wave 0's vector DMA in the iteration behind the one that issued it
    if (next_w1) {
      if (extra) ff_wait_barrier<3>();
      else ff_wait_barrier<2>();
    } else if (wave < 4) ff_wait_barrier<3>();
    else ff_wait_barrier<2>();
  };
  auto phase1 = [&](auto r_c, const unsigned char* slot, int cidx) {
    constexpr int R = decltype(r_c)::value;
    if (R == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    op16x8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xh[i] = *(const op16x8*)(xP + R * 8192 + (wm * 2 + i) * 2048 + foff_hi);
      xl[i] = *(const op16x8*)(xP + R * 8192 + (wm * 2 + i) * 2048 + foff_lo);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      wh[j] = *(const op16x8*)(slot + ((wn * 2 + j) * 2) * 1024 + lane * 16);
      wl[j] = *(const op16x8*)(slot + ((wn * 2 + j) * 2 + 1) * 1024 + lane * 16);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = FF_MFMA(wh[j], xl[i], acc1[i][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = FF_MFMA(wl[j], xh[i], acc1[i][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = FF_MFMA(wh[j], xh[i], acc1[i][j]);
    if (R == 9) {
      // ---- LayerNorm fold + bias + GEGLU on the lane's 4 consecutive features of row c, -> hT (A layout, 8-byte pieces)
      // (the vectors arrived by LDS-DMA; the counted waits cover them.  Read through inline asm: for a C++ read of an LDS-DMA
      //  destination the compiler inserts `s_waitcnt vmcnt(0)`, which would drain the two weight stages in flight once per chunk)
      const float* vb = vecs + (cidx & 1) * 256 + wn * 16 + gq * 4;
      const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)vb;
      f32x4 sv, sg, bv, bg;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\tds_read_b128 %3, %4 offset:768\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(sv), "=&v"(sg), "=&v"(bv), "=&v"(bg)
                   : "v"(vaddr)
                   : "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (wm * 2 + i) * 16 + c;
        const float mean = s_rows[row * 2], rstd = s_rows[row * 2 + 1];
        float v0 = (acc1[i][0][0] * scale1 - mean * sv[0]) * rstd + bv[0], v1 = (acc1[i][0][1] * scale1 - mean * sv[1]) * rstd + bv[1];
        float v2 = (acc1[i][0][2] * scale1 - mean * sv[2]) * rstd + bv[2], v3 = (acc1[i][0][3] * scale1 - mean * sv[3]) * rstd + bv[3];
        float g0 = (acc1[i][1][0] * scale1 - mean * sg[0]) * rstd + bg[0], g1 = (acc1[i][1][1] * scale1 - mean * sg[1]) * rstd + bg[1];
        float g2 = (acc1[i][1][2] * scale1 - mean * sg[2]) * rstd + bg[2], g3 = (acc1[i][1][3] * scale1 - mean * sg[3]) * rstd + bg[3];
        gelu_erf4(g0, g1, g2, g3);
        uint32_t h0, l0, h1, l1;
        split_op16x2(v0 * g0, v1 * g1, h0, l0);
        split_op16x2(v2 * g2, v3 * g3, h1, l1);
        const int cc = (wn & 1) * 2 + (gq >> 1);
        unsigned char* base = hT + (wn >> 1) * 8192 + ((wm * 2 + i) * 2 + (c >> 3)) * 1024 + (c & 7) * 128 + (gq & 1) * 8;
        // (inline asm for the same reason as the vector reads: a C++ store next to LDS-DMA destinations gets an `s_waitcnt vmcnt(0)`)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned a_hi = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(base + ((cc ^ fsw) << 4));
        const unsigned a_lo = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(base + (((4 + cc) ^ fsw) << 4));
        const u32x2 dh = {h0, h1}, dl = {l0, l1};
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %2, %3" ::"v"(a_hi), "v"(dh), "v"(a_lo), "v"(dl) : "memory");
      }
    }
  };
  // one stage of a chunk, R = 0 .. 13 at compile time: no index arithmetic in the loop body beyond scalar pointer adds (a lone wavefront issues
  // one instruction per ~7.5 cycles: the first version of this loop, with the stage decoded from a running counter, spent ~2 200 cycles per
  // stage on ~170 instructions for 200 - 240 cycles of MFMA)
  auto step = [&](auto r_c, int cidx) {
    constexpr int R = decltype(r_c)::value;
    const bool last_chunk = cidx == 9;
    if (R < 9) wait_next(true, R == 1 && cidx > 0 && wave == 0);
    else if (R < 13) wait_next(false, false);
    else wait_next(!last_chunk, false);
    if (R == 0 && cidx > 0 && wave == 0) issue_vecs(cidx);            // (chunk 0's went out in front of the loop)
    // stage R + 2 of the sequence
    constexpr int R2 = (R + 2) % 14;
    if (R + 2 < 14) {
      if (R2 < 10) dma_stage(w1b + (size_t)cidx * 16384 + R2 * k1s, sl2, 2);
      else dma_stage(w2b + (size_t)(cidx * 2 + ((R2 - 10) >> 1)) * k2s + ((R2 - 10) & 1) * 20480, sl2, n2);
    } else if (!last_chunk) {
      dma_stage(w1b + (size_t)(cidx + 1) * 16384 + R2 * k1s, sl2, 2);
    } else {
      dma_stage(w2x + (size_t)(R2 >> 1) * k2s + (R2 & 1) * 20480, sl2, n2);
    }
    if (R < 10) {
      phase1(r_c, sl0, cidx);
    } else {
      constexpr int Q = R - 10;
      phase2(integral_constant<int, (Q & 1)>{}, sl0, hT + (Q >> 1) * 8192 + wm4 * 2048);
    }
    unsigned char* const t_ = sl0;
    sl0 = sl1;
    sl1 = sl2;
    sl2 = t_;
  };
  auto xstep = [&](auto q_c) {
    constexpr int Q = decltype(q_c)::value;
    if (Q < 9) wait_next(false, false);
    else ff_wait_barrier<0>();
    if (Q + 2 < 10) dma_stage(w2x + (size_t)((Q + 2) >> 1) * k2s + ((Q + 2) & 1) * 20480, sl2, n2);
    phase2(integral_constant<int, (Q & 1)>{}, sl0, xP + (split * 5 + (Q >> 1)) * 8192 + wm4 * 2048);
    unsigned char* const t_ = sl0;
    sl0 = sl1;
    sl1 = sl2;
    sl2 = t_;
  };
  issue_vecs(0);
  dma_stage(w1b, sl0, 2);
  dma_stage(w1b + k1s, sl1, 2);
#pragma unroll 1
  for (int cidx = 0; cidx < 10; ++cidx) unroll_steps<0, 14>([&](auto r_c) { step(r_c, cidx); });
  unroll_steps<0, 10>([&](auto q_c) { xstep(q_c); });
  // ---- the partial output: lane holds columns 4 gq .. + 3 of row c of each 16 x 16 tile -> 16-byte stores into the split's slab
  const int m = m0 + wm4 * 16 + c;
  if (m < M) {
    float* ws = p.slabs + ((size_t)split * M + m) * d2.N;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int n = h * 160 + wn2 * 80 + j * 16 + gq * 4;
        *(float4*)(ws + n) = make_float4(acc2[h][j][0], acc2[h][j][1], acc2[h][j][2], acc2[h][j][3]);
      }
  }
}

}  // namespace

bool mvd_ffchain_launch(const mvd_gemm_desc& g1, const mvd_gemm_desc& g2, float* slabs, hipStream_t s) {
  FfParams p;
  p.g1 = g1;
  p.g2 = g2;
  p.slabs = slabs;
  hipLaunchKernelGGL(ffchain_kernel, dim3((g1.M + FF_BM - 1) / FF_BM, 2), dim3(FF_THREADS), 0, s, p);
  return true;
}
