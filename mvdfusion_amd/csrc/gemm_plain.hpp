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
// In gemm.hip: the split-K reduce kernels, among them splitk_gn_kernel -- reduce + epilogue + GroupNorm / SiLU of the output (optionally over
// its concatenation with a skip tensor) in one launch, a workgroup per (image, group) with the group's values in LDS.
// One translation unit per block tile (gemm_plain_t0.hip ... gemm_plain_t4.hip) instantiates this template for its loop variants.
#pragma once
#include "gemm_device.hpp"

namespace {

// Second launch bound = wavefronts per SIMD the register allocation must leave room for.  The plain two-buffer loop of the 8-wave tiles
// (128x128, dense A: the GEGLU / QKV projections) lives on TWO co-resident workgroups per CU (4 wavefronts per SIMD: <= 128 VGPRs) hiding each
// other's barriers and DMA waits -- at 129
// registers (round 5, after an epilogue edit) it silently dropped to one and every GEGLU / QKV projection lost 10 us; the register-pipelined
// loop of the 64x80 tile wants three workgroups (<= 168).  tests/test_cpu_oracle_and_host.py checks the tiers from the kernel descriptors.
template <int BM, int BN, int WM, int WN, int NS, int AMODE, int STAGES>
__global__ __launch_bounds__(WM * WN * 64, (BM == 128 && BN == 128 && STAGES == 2 && AMODE == MVD_A_DENSE) ? 4 : ((BM == 64 && BN == 80 && STAGES == 3) ? 3 : 1))
void gemm_kernel(GemmParams p) {
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
  constexpr int EPI_BYTES = NW * WTM * LDW * 4;
  constexpr bool RING = STAGES == 6 || STAGES == 7;    // 6 / 7 = register-pipelined loop over a DEEP ring of LDS buffers (<= 4 / <= 8)
  constexpr bool PIPE = STAGES == 3 || RING;           // 3 = register-pipelined loop (two LDS buffers)
  constexpr bool STAG = STAGES == 4;                    // 4 = staggered wave groups, three LDS buffers
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

}  // namespace
