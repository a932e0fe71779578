#include "gemm_device.hpp"

namespace {

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
template <int BM, int BN, int CM, int CN, int NS, int AMODE>
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
  // ---- weight prefetch for the launches that FOLLOW (mvd_gemm_desc.pf_items): the consumer wavefronts have no memory instruction of their own
  //      until the epilogue, and a convolution runs for 30 - 170 us with the HBM mostly idle: each consumer wave requests its share of the
  //      listed weights' 128-byte lines (one dword per line and lane; the data is dropped) so that they sit in the Infinity Cache when their
  //      GEMMs start.  The loads return into ONE register that stays reserved until the k-loop is over; nothing in the consumer's k-loop
  //      waits for vmcnt (barriers below: lgkmcnt only), so the requests cost their issue slots and nothing else.
  unsigned pf_sink = 0;
  if (d.pf_items != nullptr) {
    const int gw = (blockIdx.z * gridDim.x + blockIdx.x) * NC + wave, GW = gridDim.x * gridDim.z * NC;
    // (the table is read with SCALAR loads -- constant address space -- so that no vector-memory wait sits between two items' requests)
    const __attribute__((address_space(4))) mvd_prefetch_item* tab = (const __attribute__((address_space(4))) mvd_prefetch_item*)d.pf_items;
    for (int it = 0; it < d.pf_n; ++it) {
      const unsigned char* base = (const unsigned char*)tab[it].ptr;
      const long lines = (long)((tab[it].bytes + 127) >> 7);
      for (long l = (long)gw * 64 + lane; l < lines; l += (long)GW * 64) {
        const unsigned char* a = base + (l << 7);
        asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(a) : "memory");
      }
    }
  }
  MVD_STAMP_AT(d, wave, 1);
  wait_lgkm_and_barrier();                                            // barrier P
  MVD_STAMP_AT(d, wave, 2);
  read_frags(0, fah[0], fal[0], fbh[0], fbl[0]);
  int br = 1 % NBUF;
  auto step = [&](auto parity, int it) {
    constexpr int Pq = decltype(parity)::value;
    wait_lgkm_and_barrier();                                          // B_it: my reads of k-tile it returned; k-tile it+1 landed
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
  //  the prefetch requests are asm loads the compiler's waitcnt pass does not see: wait for them HERE, while their landing
  // register is still reserved -- a late return must not land in a VGPR the epilogue has re-used (free after a k-loop of tens of us)
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(pf_sink) : "memory");
  __syncthreads();
  tile_epilogue<BM, BN, CM, CN>(p, acc, smem, m0, n0, lane, wave, AMODE == MVD_A_DENSE && d.ln_stats != nullptr ? s_rows : nullptr);
  MVD_STAMP_AT(d, wave, 8);
}

template <int BM, int BN, int CM, int CN>
void launch_ws(GemmParams& p, hipStream_t s) {
  dim3 grid(p.tiles_n * p.tiles_m, 1, p.splits), block(512);
  const bool conv = p.d.a_mode == MVD_A_CONV3X3;
  const int ns = p.d.prec;
  if (!conv && ns == 4) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 4, MVD_A_DENSE>), grid, block, 0, s, p);
  if (!conv && ns == 3) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 3, MVD_A_DENSE>), grid, block, 0, s, p);
  if (!conv && ns == 1) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 1, MVD_A_DENSE>), grid, block, 0, s, p);
  if (conv && ns == 4) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 4, MVD_A_CONV3X3>), grid, block, 0, s, p);
  if (conv && ns == 3) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 3, MVD_A_CONV3X3>), grid, block, 0, s, p);
  if (conv && ns == 1) hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, CM, CN, 1, MVD_A_CONV3X3>), grid, block, 0, s, p);
}

}  // namespace

bool mvd_gemm_launch_ws(int tile, GemmParams& p, hipStream_t s) {
  switch (tile) {
    case 1: launch_ws<128, 128, 2, 2>(p, s); return true;
    case 2: launch_ws<128, 80, 4, 1>(p, s); return true;
    case 4: launch_ws<128, 160, 2, 2>(p, s); return true;
  }
  return false;
}
