#include "gemm_device.hpp"

namespace {

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

template <int BM, int BN, int WM, int WN>
void launch_patch(GemmParams& p, hipStream_t s, int pi) {
  dim3 grid(p.tiles_n * p.tiles_m, 1, p.splits), block(WM * WN * 64);
  const int ns = p.d.prec;
#define MVD_PATCH_CASE(NS_, PI_) \
  if (ns == NS_ && pi == PI_) hipLaunchKernelGGL((conv_patch_kernel<BM, BN, WM, WN, NS_, PI_>), grid, block, 0, s, p);
  MVD_PATCH_CASE(4, 1) MVD_PATCH_CASE(4, 2) MVD_PATCH_CASE(3, 1) MVD_PATCH_CASE(3, 2) MVD_PATCH_CASE(1, 1) MVD_PATCH_CASE(1, 2)
#undef MVD_PATCH_CASE
}

}  // namespace

bool mvd_gemm_launch_patch(int tile, GemmParams& p, hipStream_t s, int pi) {
  switch (tile) {
    case 1: launch_patch<128, 128, 2, 4>(p, s, pi); return true;
    case 2: launch_patch<128, 80, 4, 1>(p, s, pi); return true;
    case 4: launch_patch<128, 160, 4, 2>(p, s, pi); return true;
  }
  return false;
}
