// gemm_pt_kernel: PERSISTENT, role-split GEMM / implicit-GEMM convolution on 128x128 output tiles (mvd_gemm_desc.cfg loop 10).
//
// Why (profiles/r03_geglu_stamp.log, r03_ws_stamp_after.log): in the tile-per-workgroup kernels of gemm.hip a launch is
// prologue -> k-loop -> epilogue in series on every CU.  For the short-K projections of the UNet (GEGLU / QKV: 10 - 40 k-tiles) half of a
// tile's life is outside its k-loop -- 4 k cycles from launch to the first MFMA, ~10 k cycles of epilogue (erf-GELU, operand split, stores)
// -- and nothing overlaps it: the matrix pipe idles while the VALU works and vice versa.  Here ONE workgroup per CU stays resident, walks a
// contiguous run of output tiles, and its sixteen wavefronts have three roles that never wait for each other except through data:
//   * 8 CONSUMERS (2 per SIMD, 64x32 wave tiles): fragment reads + MFMAs, nothing else.  At the end of a tile a consumer dumps its
//     accumulators into the LDS staging tile and goes straight into the next tile's k-loop, whose first k-tiles are already in the ring.
//   * 2 LOADERS: every LDS-DMA of the ring of 4 stages (32 KiB each), issued as soon as a stage is free, ACROSS tile boundaries -- the
//     prologue latency (TLB walk + first cold weight line) is paid once per workgroup, not once per tile.  A loader never blocks on its
//     DMA queue: it reads its own VM_CNT from HW_REG_IB_STS (profiles/r05_hw_facts_probe.log) and publishes "k-tile landed" the moment the
//     counter says so, while it keeps issuing.  DMAs are issued through inline asm in the scalar-base form (one s_mov m0 + one
//     global_load_lds per KiB: the builtin's per-lane 64-bit pointer arithmetic made a loader instruction-issue bound).
//   * 6 EPILOGUE waves: take 16-row units of the staging tile by ticket, read them into registers (which releases the staging slots to
//     the loaders within a few hundred cycles), then run bias / activation / residual / GEGLU / QKV routing / LayerNorm fold / operand
//     split and the global stores on the VALU while the consumers' MFMAs of the NEXT tile run on the matrix pipe (separate pipes).  One
//     wavefront issues a VALU instruction only every ~7.5 cycles (probe log), so the VALU side needs many wavefronts: six here.
// Synchronisation is by monotonic counters in LDS (ds_add_u32 behind the data accesses of the same wavefront: LDS executes a wavefront's
// operations in order), polled with ds_read_b32: no s_barrier after the start, so a role is never parked behind another role's tail.
// The staging tile is not extra LDS: it is the two ring stages the tile's last two k-tiles occupied, handed from the consumers to the
// epilogue waves and from those back to the loaders.
// Numerics: same k order, same per-accumulator product order (lo*lo, lo*hi, hi*lo, hi*hi) and the same epilogue arithmetic as
// gemm_kernel => bit-identical outputs (the MFMA operands are swapped so that a lane holds four consecutive COLUMNS of a row;
// v_mfma_f32_16x16x32_f16(b, a) is the bitwise transpose of (a, b): probe log).  Row / GroupNorm statistics are summed in another order
// than gemm_kernel's (documented in mvd_hip.h: the slots / partials are not part of the bit-exact contract).
#include <type_traits>

#include "gemm_common.hpp"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_pt_zero_page[4];

constexpr int NCW = 8, NLW = 4, NEW = 4;               // consumer / loader / epilogue wavefronts
constexpr int PT_THREADS = (NCW + NLW + NEW) * 64;
constexpr int NBUF = 4, STAGE = 32 * 1024;
constexpr int LGR = 16;                                // granules (1 KiB) of each operand per k-tile
constexpr int RING_BYTES = NBUF * STAGE;
constexpr int OFF_TAB = RING_BYTES;                    // conv: per loader, source offset of (tile row) x (tap)
constexpr int TAB_BYTES = NLW * 128 * 9 * 4;
constexpr int OFF_ECOL = OFF_TAB + TAB_BYTES;          // per epilogue wave: column {sum, sum of squares} [128][2]
constexpr int OFF_FLAGS = OFF_ECOL + NEW * 1024;
enum { F_FULL = 0, F_CREAD = 4, F_ECNT = 8, F_STAG = 12, F_TICKET = 14, F_ABORT = 15, F_COUNT = 16 };
constexpr int PT_SMEM = OFF_FLAGS + F_COUNT * 4;
static_assert(PT_SMEM <= 160 * 1024, "LDS budget");
constexpr int SPIN_MAX = 1 << 20;

// -DMVD_PT_STAMP (tools/probes/pt_stamp.sh): cycle accounting of workgroup 0 -- consumer 0, loader 0 and epilogue wave 0 add up where their
// time goes and write 16 int64 each into d.workspace (which the stamped problems do not use: no split-K)
// -DMVD_PT_VARIANT=bits (probe builds only; results are garbage): 1 = consumers skip their MFMAs, 2 = consumers skip their fragment reads,
// 4 = loaders issue no DMA (publish at once), 8 = epilogue waves release the staging tile and do nothing else
#ifndef MVD_PT_VARIANT
#define MVD_PT_VARIANT 0
#endif
#ifdef MVD_PT_STAMP
#define PT_NOW() ((long long)__builtin_readcyclecounter())
#define PT_T(var) const long long var = PT_NOW()
#define PT_ACC(acc, t0) acc += PT_NOW() - (t0)
#define PT_DUMP(d, role, ...)                                                                   \
  do {                                                                                          \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                           \
      const long long vals_[] = {__VA_ARGS__};                                                  \
      for (unsigned i_ = 0; i_ < sizeof(vals_) / 8; ++i_) ((long long*)(d).workspace)[(role) * 16 + i_] = vals_[i_]; \
    }                                                                                           \
  } while (0)
#else
#define PT_NOW() 0ll
#define PT_T(var) do {} while (0)
#define PT_ACC(acc, t0) do {} while (0)
#define PT_DUMP(d, role, ...) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------- LDS counters
__device__ __forceinline__ unsigned lds_ld(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// two counters 16 bytes apart with one instruction
__device__ __forceinline__ void lds_ld2(unsigned addr, unsigned& a, unsigned& b) {
  unsigned long long v;
  asm volatile("ds_read2_b32 %0, %1 offset1:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  b = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
}
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// DS atomics are per LANE: with all 64 lanes active a wavefront adds 64 -- and the LDS serialises the 64 same-address updates (~1 400 cycles
// per ds_add + read round trip with sixteen wavefronts doing it: the whole kernel ran at the pace of its counters, whatever the MFMA / DMA
// load -- profiles/r05_pt_variants_v4.log).  The adds therefore run with exec = lane 0 only, set and restored inside the asm statement.
__device__ __forceinline__ void lds_add1(unsigned addr) {
  unsigned long long keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "v"(addr), "v"(1u) : "memory");
}
__device__ __forceinline__ unsigned lds_ticket(unsigned addr) {
  unsigned v;
  unsigned long long keep;
  asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v), "=&s"(keep)
               : "v"(addr), "v"(1u)
               : "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ bool ge(unsigned v, unsigned events) { return (int)(v - events) >= 0; }
// spin until the counter at `addr` reaches `need`; bounded: a protocol bug ends the kernel with poisoned outputs instead of hanging the box
__device__ __forceinline__ bool wait_ge(unsigned addr, unsigned need, unsigned abort_addr) {
  if (ge(lds_ld(addr), need)) return true;
  for (int i = 0; i < SPIN_MAX; ++i) {
    __builtin_amdgcn_s_sleep(1);
    if (ge(lds_ld(addr), need)) return true;
    if ((i & 255) == 255 && lds_ld(abort_addr) != 0) return false;
  }
  lds_st(abort_addr, 1);
  return false;
}

__device__ __forceinline__ unsigned read_vmcnt() {      // HW_REG_IB_STS: VM_CNT = bits [3:0] | bits [23:22] << 4
  const unsigned v = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 7);
  return (v & 15u) | (((v >> 22) & 3u) << 4);
}

// ---------------------------------------------------------------------------------------------- work partition
// Items = (split z, output tile), z-major; the 8 XCDs take contiguous runs (workgroup b runs on XCD b % 8), the workgroups of an XCD
// contiguous sub-runs: consecutive tiles of a workgroup share an A row panel (n-fastest) or a W column panel (m-fastest) in its XCD's L2.
__device__ __forceinline__ void pt_my_items(const GemmParams& p, int& i0, int& i1) {
  const int T = p.tiles_m * p.tiles_n * p.splits, G = gridDim.x, b = blockIdx.x;
  const int x = b & 7, j = b >> 3;
  const int J = (G + 7 - x) >> 3;
  const int q = T >> 3, r = T & 7;
  const int s = x * q + (x < r ? x : r), cnt = q + (x < r ? 1 : 0);
  i0 = s + (int)((long)j * cnt / J);
  i1 = s + (int)((long)(j + 1) * cnt / J);
}
struct PtItem {
  int m0, n0, z, kt0, nkt;
};
__device__ __forceinline__ PtItem pt_item(const GemmParams& p, int item) {
  const int tiles = p.tiles_m * p.tiles_n;
  PtItem it;
  it.z = item / tiles;
  const int t = item - it.z * tiles;
  it.m0 = (p.m_fastest ? t % p.tiles_m : t / p.tiles_n) * 128;
  it.n0 = (p.m_fastest ? t / p.tiles_m : t % p.tiles_n) * 128;
  it.kt0 = it.z * p.kt_per_split;
  it.nkt = min(p.nk, it.kt0 + p.kt_per_split) - it.kt0;
  return it;
}
__device__ __forceinline__ int pt_item_nkt(const GemmParams& p, int item) {
  const int z = item / (p.tiles_m * p.tiles_n);
  return min(p.nk, (z + 1) * p.kt_per_split) - z * p.kt_per_split;
}

// ---------------------------------------------------------------------------------------------- LDS-DMA (inline asm: hipcc does not count these)
// Sixteen 1 KiB granules to LDS dst, dst + 1 KiB, ...  The immediate offset of global_load_lds moves the LDS destination AND the global
// source (tools/probes/dma_offset_probe.hip): M0 is written once per FOUR granules and the offsets 0 / 1 / 2 / 3 KiB do the rest -- 1.5
// instructions per KiB instead of 3 (a single wavefront issues one instruction per ~8 cycles: the loaders are issue bound).  The callers
// pre-compensate the sources: granule g's lane offset is v[g] = true offset + 3072 - (g & 3) * 1024 against sbase - 3072.
__device__ __forceinline__ void dma16_saddr(const unsigned* v, const void* sbase, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %18\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %17\n\tglobal_load_lds_dwordx4 %2, %17 offset:1024\n\t"
      "global_load_lds_dwordx4 %3, %17 offset:2048\n\tglobal_load_lds_dwordx4 %4, %17 offset:3072\n\t"
      "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %17\n\tglobal_load_lds_dwordx4 %6, %17 offset:1024\n\t"
      "global_load_lds_dwordx4 %7, %17 offset:2048\n\tglobal_load_lds_dwordx4 %8, %17 offset:3072\n\t"
      "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %9, %17\n\tglobal_load_lds_dwordx4 %10, %17 offset:1024\n\t"
      "global_load_lds_dwordx4 %11, %17 offset:2048\n\tglobal_load_lds_dwordx4 %12, %17 offset:3072\n\t"
      "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %13, %17\n\tglobal_load_lds_dwordx4 %14, %17 offset:1024\n\t"
      "global_load_lds_dwordx4 %15, %17 offset:2048\n\tglobal_load_lds_dwordx4 %16, %17 offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]),
        "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "s"(sbase), "s"(dst)
      : "memory", "scc");
}
// ... sources as per-lane 64-bit pointers, pre-compensated by -(g & 3) * 1024 (conv A: a tap in the zero padding sources the zero page)
__device__ __forceinline__ void dma16_vaddr(const void* const* a, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %17\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
      "global_load_lds_dwordx4 %3, off offset:2048\n\tglobal_load_lds_dwordx4 %4, off offset:3072\n\t"
      "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %5, off\n\tglobal_load_lds_dwordx4 %6, off offset:1024\n\t"
      "global_load_lds_dwordx4 %7, off offset:2048\n\tglobal_load_lds_dwordx4 %8, off offset:3072\n\t"
      "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %9, off\n\tglobal_load_lds_dwordx4 %10, off offset:1024\n\t"
      "global_load_lds_dwordx4 %11, off offset:2048\n\tglobal_load_lds_dwordx4 %12, off offset:3072\n\t"
      "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %13, off\n\tglobal_load_lds_dwordx4 %14, off offset:1024\n\t"
      "global_load_lds_dwordx4 %15, off offset:2048\n\tglobal_load_lds_dwordx4 %16, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
        "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "s"(dst)
      : "memory", "scc");
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

// ============================================================================================== LOADER
// FOUR loaders, loader l OWNS ring slot l: it stages the k-tiles q = l, l + 4, l + 8, ... whole (16 A + 16 B granules).  Why four and why
// whole k-tiles (profiles/r05_pt_stamp_v*.log): a single wavefront issues ONE instruction per ~8 cycles, and the bookkeeping of a k-tile --
// counters, item cursor, addresses: ~150 scalar instructions as hipcc compiles it -- costs a loader ~1 200 cycles whatever the number of DMAs
// behind it; two loaders that each touch every k-tile deliver one k-tile per ~1 500 cycles (MFMA time of the tile: 768).  With four, a loader
// has four k-tile times per k-tile of its own.  Its VM_CNT (6 bits) holds one whole k-tile (32 DMAs) comfortably.
template <int AMODE>
__device__ __forceinline__ void pt_loader(const GemmParams& p, unsigned char* smem, unsigned lds0, int i0, int i1, int l, int lane) {
  static_assert(NLW == NBUF, "a loader owns one ring slot");
  const mvd_gemm_desc& d = p.d;
  const unsigned fl = lds0 + OFF_FLAGS, abort_addr = fl + F_ABORT * 4;
  int* const tab = (int*)(smem + OFF_TAB) + l * 128 * 9;
  const int gr = lane >> 3;
  // Empty the compiler's VMEM scoreboard here: a register reloaded from scratch in the common prologue would otherwise get its
  // `s_waitcnt vmcnt(0)` at its first use INSIDE the issue loop below (seen in the ISA) -- and there it waits for every LDS-DMA in flight.
  __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0) only
  int Qtotal = 0;
  for (int i = i0; i < i1; ++i) Qtotal += pt_item_nkt(p, i);

  // cursor over ALL k-tiles of the workgroup (the other loaders' are skipped, but every item's geometry is walked)
  int item = i0, it = 0, nkt = 0, kt0 = 0, tab_m0 = -1;
  unsigned voffA[LGR], voffB[LGR];
  int a_chunk[LGR];                                        // conv: chunk offset of this lane inside granule g's 128-byte line
  const size_t b_kbytes = d.b_mode == MVD_B_PLANES ? (size_t)128 : (size_t)p.nt16 * 2048;
  auto setup_item = [&]() {
    const PtItem w = pt_item(p, item);
    nkt = w.nkt;
    kt0 = w.kt0;
    it = 0;
#pragma unroll
    for (int gi = 0; gi < LGR; ++gi) {                     // granule = 8-row group of the tile (A) / fragment image or 8-row group (B)
      const int R = (gi & 1) * 8 + gr;
      const int gc = (lane & 7) ^ ((R >> 1) & 7);
      const unsigned comp = 3072u - (unsigned)(gi & 3) * 1024u;       // (dma16_saddr: immediate offsets 0 .. 3 KiB against a base 3 KiB lower)
      int m = w.m0 + gi * 8 + gr;
      if (m > d.M - 1) m = d.M - 1;                        // rows past M: any valid line (the epilogue never stores them)
      voffA[gi] = (unsigned)m * (unsigned)(4 * d.lda) + gc * 16 + comp;
      a_chunk[gi] = gc * 8;
      if (d.b_mode == MVD_B_PLANES) {
        int n = w.n0 + gi * 8 + gr;
        if (n > d.N - 1) n = d.N - 1;
        voffB[gi] = (unsigned)n * (unsigned)(4 * d.ldb) + gc * 16 + comp;
      } else {
        int nt = (w.n0 >> 4) + (gi >> 1);
        if (nt > p.nt16 - 1) nt = p.nt16 - 1;
        voffB[gi] = (unsigned)nt * 2048u + (gi & 1) * 1024 + lane * 16 + comp;
      }
    }
    if (AMODE != MVD_A_DENSE && w.m0 != tab_m0) {          // source offset of every (tile row, tap); -1: zero padding / past M
      tab_m0 = w.m0;
      const int hw = d.Hout * d.Wout;
      for (int e = lane; e < 128 * 9; e += 64) {
        const int row = e / 9, tap = e - row * 9;
        const int m = w.m0 + row;
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
        tab[e] = off;
      }
    }
  };
  const unsigned dstA = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + l * STAGE)), dstB = dstA + 16384;
  auto issue_a = [&]() {
    if (AMODE == MVD_A_DENSE) {
      dma16_saddr(voffA, uniform_ptr((const unsigned char*)d.A + (size_t)(kt0 + it) * 128 - 3072), dstA);
    } else {
      const int kt = kt0 + it, cb = kt / 9, tap = kt - cb * 9;
      const void* src[LGR];
#pragma unroll
      for (int gi = 0; gi < LGR; ++gi) {
        const int off = tab[(gi * 8 + gr) * 9 + tap];
        const unsigned char* sp = off >= 0 ? (const unsigned char*)((const u16*)d.A + (unsigned)(off + cb * 64 + a_chunk[gi]))
                                           : (const unsigned char*)g_pt_zero_page;
        src[gi] = sp - (gi & 3) * 1024;
      }
      dma16_vaddr(src, dstA);
    }
  };
  auto issue_b = [&]() { dma16_saddr(voffB, uniform_ptr((const unsigned char*)d.Wp + (size_t)(kt0 + it) * b_kbytes - 3072), dstB); };

  // My slot's latest use may have ended as (half of) a staging tile: then the EPILOGUE waves release it (F_ECNT: four 16-row units per
  // staged use), else the consumers' fragment reads do (F_CREAD counts every use).
  unsigned uses = 0, staged_uses = 0;   // uses of my slot issued so far / of which staged
  bool staged_last = false;
  int qi = l;                           // my next k-tile (global index)
  int passed = 0;                       // k-tiles the cursor (item, it) has walked over
  int mine_signalled = 0;
  const int mine_total = Qtotal > l ? (Qtotal - l + NLW - 1) / NLW : 0;
  int idle = 0;
  [[maybe_unused]] long long st_iter = 0, st_idle = 0, st_issue = 0, st_notfree = 0, st_flyfull = 0;
  [[maybe_unused]] const long long st_t0 = PT_NOW();
  auto walk_to = [&](int q) {           // move the cursor to k-tile q (whole items at a time where possible)
    while (passed < q) {
      if (it == nkt) setup_item();
      const int step = min(nkt - it, q - passed);
      it += step;
      passed += step;
      if (it == nkt && passed < Qtotal) ++item;
      else if (it == nkt) break;
    }
    if (it == nkt && passed < Qtotal) setup_item();
  };
  while (mine_signalled < mine_total) {
    bool progress = false;
#ifdef MVD_PT_STAMP
    ++st_iter;
#endif
    // ---- has my k-tile in flight landed?  (at most one: the next use of my slot needs it consumed first)
    if ((int)uses > mine_signalled) {
      const bool landed = (MVD_PT_VARIANT & 4) ? true : read_vmcnt() == 0;
      if (landed) {
        lds_add1(fl + (F_FULL + l) * 4);
        ++mine_signalled;
        progress = true;
      }
    }
    // ---- issue my next k-tile as soon as my slot is free
    if (qi < Qtotal && (int)uses == mine_signalled) {
      bool free_ = true;
      if (uses > 0) {
        unsigned cr, ec;
        lds_ld2(fl + (F_CREAD + l) * 4, cr, ec);
        free_ = ge(cr, NCW * uses);                                         // every consumer has read all earlier uses of the slot ...
        if (staged_last) free_ = free_ && ge(ec, 4u * staged_uses);         // ... and the epilogue waves the staging units it last held
      }
#ifdef MVD_PT_STAMP
      if (!free_) ++st_notfree;
#endif
      if (free_) {
        PT_T(ti);
        walk_to(qi);
        if (!(MVD_PT_VARIANT & 4)) {
          issue_a();
          issue_b();
        }
        PT_ACC(st_issue, ti);
        staged_last = it >= nkt - 2;                                        // the tile's last two k-tiles: their slots become its staging tile
        if (staged_last) ++staged_uses;
        ++uses;
        qi += NLW;
        progress = true;
      }
    }
#ifdef MVD_PT_STAMP
    if (!progress) ++st_idle;
#endif
    if (!progress) {
      __builtin_amdgcn_s_sleep(1);
      if (((++idle) & 1023) == 0 && lds_ld(abort_addr) != 0) return;
      if (idle > (SPIN_MAX << 2)) {
        lds_st(abort_addr, 1);
        return;
      }
    } else {
      idle = 0;
    }
  }
  if (l == 0) PT_DUMP(d, 1, PT_NOW() - st_t0, st_iter, st_idle, st_issue, st_notfree, st_flyfull, (long long)mine_total);
}

// ============================================================================================== CONSUMER
template <int NS>
__device__ __forceinline__ void pt_consumer(const GemmParams& p, unsigned char* smem, unsigned lds0, int i0, int i1, int w, int lane) {
  const mvd_gemm_desc& d = p.d;
  const unsigned fl = lds0 + OFF_FLAGS, abort_addr = fl + F_ABORT * 4;
  const int wm = w >> 2, wn = w & 3;                       // 2 x 4 wave tiles of 64 x 32
  constexpr int TM = 4, TN = 2;
  const int frow = lane & 15;
  const int fsw = (frow >> 1) & 7;
  const int fbase = (frow >> 3) * 1024 + (frow & 7) * 128;
  const int foff_hi = fbase + (((lane >> 4)) ^ fsw) * 16;
  const int foff_lo = fbase + ((4 + (lane >> 4)) ^ fsw) * 16;
  const int boff_hi = d.b_mode == MVD_B_PLANES ? foff_hi : lane * 16;
  const int boff_lo = d.b_mode == MVD_B_PLANES ? foff_lo : 1024 + lane * 16;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  unsigned q = 0, Qtotal = 0;
  for (int i = i0; i < i1; ++i) Qtotal += pt_item_nkt(p, i);
  [[maybe_unused]] long long st_full = 0, st_dumpw = 0, st_dump = 0, st_nwait = 0;
  [[maybe_unused]] const long long st_t0 = PT_NOW();
  if (!wait_ge(fl + F_FULL * 4, 1, abort_addr)) return;
  [[maybe_unused]] const long long st_t1 = PT_NOW();
  // Two nested loops on purpose: with ONE flat loop over k-tiles and the tile end (dump, accumulators = 0) as a branch inside it, the
  // accumulators become a phi of {MFMA result, zero} and hipcc copies all 32 of them to other registers and back in EVERY iteration
  // (62 v_mov behind a drained matrix pipe: 2 300 cycles per k-tile instead of 768 -- profiles/r05_pt_stamp_v1.log).
  for (int item = i0; item < i1; ++item) {
    const int nkt = pt_item_nkt(p, item);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {
      const int slot = q & 3;
      const unsigned char* sA = smem + slot * STAGE;
      const unsigned char* sB = sA + 16384;
      op16x8 ah[TM], al[TM], bh[TN], bl[TN];
      // ALL fragment reads first, in the order the MFMAs want them (lo A / hi B feed the first product term): left to itself hipcc issues
      // each read right before the MFMA pair that needs it, behind `lgkmcnt` waits -- six exposed LDS round trips per k-tile, 2 200 cycles
      // per k-tile with two consumer waves per SIMD (profiles/r05_pt_stamp_v2.log).  The scheduling barrier below pins reads | MFMAs; the
      // compiler's counted lgkmcnt waits then let the first MFMAs start while the later reads are still in flight.
#if MVD_PT_VARIANT & 2
#pragma unroll
      for (int i = 0; i < TM; ++i) ah[i] = al[i] = (op16x8){};
#pragma unroll
      for (int j = 0; j < TN; ++j) bh[j] = bl[j] = (op16x8){};
      if (false)
#endif
      {
      if (NS >= 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) al[i] = *(const op16x8*)(sA + (wm * TM + i) * 2048 + foff_lo);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) bh[j] = *(const op16x8*)(sB + (wn * TN + j) * 2048 + boff_hi);
      if (NS >= 3) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bl[j] = *(const op16x8*)(sB + (wn * TN + j) * 2048 + boff_lo);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) ah[i] = *(const op16x8*)(sA + (wm * TM + i) * 2048 + foff_hi);
      }
      // behind the fragment reads in this wave's LDS queue: "I have read this slot" (LDS executes a wave's operations in order) and a
      // first look at the NEXT k-tile's landed counter -- its value is inspected after the MFMAs, by when it has long returned
      lds_add1(fl + (F_CREAD + slot) * 4);
      unsigned fnext;
      asm volatile("ds_read_b32 %0, %1" : "=v"(fnext) : "v"(fl + (F_FULL + ((q + 1) & 3)) * 4) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      // operands swapped: D = B A^T, a lane holds row (lane & 15) of the 16-row block and columns (lane >> 4) * 4 .. + 3 of the 16-column
      // block.  Term-major order; every accumulator receives lo*lo, lo*hi, hi*lo, hi*hi in that order per k-tile (as gemm_kernel).
#if MVD_PT_VARIANT & 1
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
      if (false)
#endif
      {
      if (NS == 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(bl[j], al[i], acc[i][j], 0, 0, 0);
      }
      if (NS >= 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(bl[j], ah[i], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MVD_MFMA_16x16x32(bh[j], ah[i], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);                     // (keeps the wait below BEHIND the MFMAs: it would drain the fragment reads early)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fnext)::"memory");
      ++q;
      if (q < Qtotal) {                                      // the next k-tile (of this tile or of the next one) must have landed
        const unsigned need = (q >> 2) + 1;                   // (the slot's owner publishes the whole k-tile)
        if (!ge((unsigned)__builtin_amdgcn_readfirstlane((int)fnext), need)) {
          PT_T(tw);
          if (!wait_ge(fl + (F_FULL + (q & 3)) * 4, need, abort_addr)) return;
          PT_ACC(st_full, tw);
#ifdef MVD_PT_STAMP
          ++st_nwait;
#endif
        }
      }
    }
    // ---- tile done: accumulators -> staging tile = the slots of this tile's last two k-tiles (rows 0-63 in the older one, 64-127 in the
    //      newer), once every consumer has read its fragments out of my half's slot
    const unsigned qs = wm == 0 ? q - 2 : q - 1;             // the k-tile whose slot my rows go to
    const int sslot = qs & 3;
    PT_T(td0);
    if (!wait_ge(fl + (F_CREAD + sslot) * 4, NCW * ((qs >> 2) + 1), abort_addr)) return;
    PT_ACC(st_dumpw, td0);
    PT_T(td1);
    unsigned char* st = smem + sslot * STAGE;
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) *(f32x4*)(st + (i * 16 + r16) * 512 + (((wn * 8 + j * 4 + g) ^ r16) << 4)) = acc[i][j];
    asm volatile("" ::: "memory");
    lds_add1(fl + (F_STAG + wm) * 4);
    PT_ACC(st_dump, td1);
  }
  if (w == 0) PT_DUMP(d, 0, PT_NOW() - st_t0, st_t1 - st_t0, st_full, st_nwait, st_dumpw, st_dump, (long long)q);
}

// ============================================================================================== EPILOGUE waves
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }

// rows of a 16-row unit: mean / rstd of the folded LayerNorm, lane r (< 16) = row r of the unit (gathered while the tile is still being multiplied)
__device__ __forceinline__ float2 pt_ln_row(const mvd_gemm_desc& d, int m) {
  const int cnt = d.ln_count[0];
  const float2* p = (const float2*)d.ln_stats + (size_t)m * d.ln_ld;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < cnt; i += 4) {
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

__device__ __forceinline__ float pt_act(float v, int act) {
  if (act == MVD_ACT_GELU) return gelu_erf(v);
  if (act == MVD_ACT_SILU) return silu_f(v);
  if (act == MVD_ACT_QUICKGELU) return v / (1.0f + expf(-1.702f * v));
  return v;
}

struct PtUnit {
  const unsigned char* st;     // staging slot of the unit's half
  int rbase;                   // first row of the unit inside its half (0, 16, 32, 48)
  int m0u, n0;                 // global row of the unit's first row, first column of the tile
  int z;
};

// float4 (row, 16-byte chunk) of the staging half
__device__ __forceinline__ float4 st_ld4(const unsigned char* st, int row, int chunk) {
  return *(const float4*)(st + row * 512 + ((chunk ^ (row & 15)) << 4));
}

// ---- MVD_EPI_STORE (and the split-K slab).  Lane = (row pair member lane >> 5, 16-byte column chunk lane & 31); t = 0..7: row t * 2 + (lane >> 5).
template <int ACT, bool HAS_RES, bool HAS_BB>
__device__ __forceinline__ void pt_epi_store(const GemmParams& p, const PtUnit& u, int lane, float* ecol, unsigned ecnt_addr) {
  const mvd_gemm_desc& d = p.d;
  const int chunk = lane & 31, rsel = lane >> 5;
  const int n = u.n0 + chunk * 4;
  float4 v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = st_ld4(u.st, u.rbase + t * 2 + rsel, chunk);
  lds_add1(ecnt_addr);                                      // (behind the reads: this unit of the staging tile is free)
  const bool col_ok = n < d.n_store;
  const int mrow = u.m0u + rsel;
  const float* const zero = (const float*)g_pt_zero_page;
  const float scale = gemm_acc_scale(d);
  const bool has_bias = d.bias != nullptr, has_cs = d.colscale != nullptr;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(1.f, 1.f, 1.f, 1.f);
  if (has_bias && col_ok) b = ld4(d.bias + n);
  if (has_cs && col_ok) cs = ld4(d.colscale + n);
  const bool use_res = HAS_RES && d.res != nullptr, use_bb = HAS_BB && d.bias_b != nullptr;
  const int rpb = use_bb ? d.rows_per_batch : 1;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float4 qr[4], qb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = h * 4 + j;
      const int m = mrow + t * 2;
      const bool ok = col_ok && m < d.M;
      if (HAS_RES) qr[j] = ld4(use_res && ok ? d.res + (size_t)m * d.ldr + n : zero);
      if (HAS_BB) qb[j] = ld4(use_bb && ok ? d.bias_b + (size_t)(m / rpb) * d.ldbb + n : zero);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = h * 4 + j;
      float4 x = v[t];
      x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale;
      if (has_bias) {
        x.x += b.x; x.y += b.y; x.z += b.z; x.w += b.w;
      }
      if (HAS_BB) {
        x.x += qb[j].x; x.y += qb[j].y; x.z += qb[j].z; x.w += qb[j].w;
      }
      if (ACT == MVD_ACT_GELU) {
        gelu_erf4(x.x, x.y, x.z, x.w);
      } else if (ACT != MVD_ACT_NONE) {
        x.x = pt_act(x.x, ACT); x.y = pt_act(x.y, ACT); x.z = pt_act(x.z, ACT); x.w = pt_act(x.w, ACT);
      }
      if (has_cs) {
        x.x *= cs.x; x.y *= cs.y; x.z *= cs.z; x.w *= cs.w;
      }
      if (HAS_RES) {
        x.x += qr[j].x; x.y += qr[j].y; x.z += qr[j].z; x.w += qr[j].w;
      }
      v[t] = x;
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int m = mrow + t * 2;
    if (col_ok && m < d.M) {
      if (d.out) *(float4*)(d.out + (size_t)m * d.ldo + n) = v[t];
      if (d.out_sp) store_sp4((u16*)d.out_sp, (size_t)m, d.ldp, n, v[t].x, v[t].y, v[t].z, v[t].w);
    }
  }
  if (d.rs_out) {
    // per-row {sum, sum of squares} over the tile's 128 columns -> slot n0 / 128 of the row (the consumer adds the slots in order)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int m = mrow + t * 2;
      float s1 = 0.f, q1 = 0.f;
      if (col_ok) {
        s1 = (v[t].x + v[t].y) + (v[t].z + v[t].w);
        q1 = (v[t].x * v[t].x + v[t].y * v[t].y) + (v[t].z * v[t].z + v[t].w * v[t].w);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        q1 += __shfl_xor(q1, o, 64);
      }
      if (chunk == 0 && m < d.M) *((float2*)d.rs_out + (size_t)m * d.rs_ld + (u.n0 >> 7)) = make_float2(s1, q1);
    }
    if (u.m0u == 0 && u.n0 == 0 && lane == 0) d.rs_count[0] = (d.n_store + 127) >> 7;
  }
  if (d.gn_stats) {
    // GroupNorm statistics of the output: the unit's 16 rows lie in one image (gn_hw % 16 == 0).  Column sums over the rows (this lane's 8,
    // then the partner half), parked in this wave's LDS strip; the first lane of every group fragment adds its columns in order.
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (mrow + t * 2 < d.M) {
        s.x += v[t].x; s.y += v[t].y; s.z += v[t].z; s.w += v[t].w;
        q.x += v[t].x * v[t].x; q.y += v[t].y * v[t].y; q.z += v[t].z * v[t].z; q.w += v[t].w * v[t].w;
      }
    }
    s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64); s.z += __shfl_xor(s.z, 32, 64); s.w += __shfl_xor(s.w, 32, 64);
    q.x += __shfl_xor(q.x, 32, 64); q.y += __shfl_xor(q.y, 32, 64); q.z += __shfl_xor(q.z, 32, 64); q.w += __shfl_xor(q.w, 32, 64);
    if (rsel == 0) {
      *(float4*)(ecol + chunk * 4) = s;
      *(float4*)(ecol + 128 + chunk * 4) = q;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cg = d.n_store / d.gn_groups;
#pragma unroll 1
    for (int c0 = 0; c0 < 128; c0 += 64) {
      const int col = c0 + lane, nn = u.n0 + col;
      const bool okc = nn < d.n_store;
      const int gidx = okc ? nn / cg : 0, pos = okc ? nn - gidx * cg : 0;
      if (okc && (pos == 0 || col == 0)) {
        int len = cg - pos;
        if (len > 128 - col) len = 128 - col;
        if (len > d.n_store - nn) len = d.n_store - nn;
        float ss = 0.f, qq = 0.f;
        for (int j = 0; j < len; ++j) {
          ss += ecol[col + j];
          qq += ecol[128 + col + j];
        }
        gn_stats_add(d.gn_stats, u.m0u / d.gn_hw, gidx, d.gn_groups, ss, qq);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ __forceinline__ void pt_epi_slab(const GemmParams& p, const PtUnit& u, int lane, unsigned ecnt_addr) {
  const mvd_gemm_desc& d = p.d;
  const int chunk = lane & 31, rsel = lane >> 5;
  const int n = u.n0 + chunk * 4;
  float4 v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = st_ld4(u.st, u.rbase + t * 2 + rsel, chunk);
  lds_add1(ecnt_addr);
  float* ws = d.workspace + (size_t)u.z * d.M * d.N;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int m = u.m0u + t * 2 + rsel;
    if (m < d.M && n < d.N) *(float4*)(ws + (size_t)m * d.N + n) = v[t];
  }
}

// ---- MVD_EPI_GEGLU: packed column block of 32 = 16 value | 16 gate.  Lane = (row member lane >> 4, pair lane & 15: block (lane & 15) >> 2, 16-byte
//      sub-chunk lane & 3); t = 0..3: row t * 4 + (lane >> 4).
template <bool LNF>
__device__ __forceinline__ void pt_epi_geglu(const GemmParams& p, const PtUnit& u, int lane, float2 lnrow, unsigned ecnt_addr) {
  const mvd_gemm_desc& d = p.d;
  const int pr = lane & 15, rsel = lane >> 4;
  const int blk = pr >> 2, sub = pr & 3;
  float4 vv[4], gg[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    vv[t] = st_ld4(u.st, u.rbase + t * 4 + rsel, blk * 8 + sub);
    gg[t] = st_ld4(u.st, u.rbase + t * 4 + rsel, blk * 8 + 4 + sub);
  }
  lds_add1(ecnt_addr);
  const int nblk = u.n0 + blk * 32;                         // packed column of the block
  const bool col_ok = nblk < d.N;
  const int col = (nblk >> 5) * 16 + sub * 4;               // output column
  const int half = d.N >> 1;
  float4 sv = make_float4(0.f, 0.f, 0.f, 0.f), sg = sv, bv = sv, bg = sv;
  if (LNF && col_ok) {
    sv = ld4(d.ln_colsum + col);
    sg = ld4(d.ln_colsum + half + col);
  }
  if (d.bias && col_ok) {
    bv = ld4(d.bias + col);
    bg = ld4(d.bias + half + col);
  }
  const float scale = gemm_acc_scale(d);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float4 v = vv[t], g = gg[t];
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
    if (LNF) {
      const float mean = __shfl(lnrow.x, t * 4 + rsel, 64), rstd = __shfl(lnrow.y, t * 4 + rsel, 64);
      v.x = (v.x - mean * sv.x) * rstd; v.y = (v.y - mean * sv.y) * rstd; v.z = (v.z - mean * sv.z) * rstd; v.w = (v.w - mean * sv.w) * rstd;
      g.x = (g.x - mean * sg.x) * rstd; g.y = (g.y - mean * sg.y) * rstd; g.z = (g.z - mean * sg.z) * rstd; g.w = (g.w - mean * sg.w) * rstd;
    }
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    g.x += bg.x; g.y += bg.y; g.z += bg.z; g.w += bg.w;
    gelu_erf4(g.x, g.y, g.z, g.w);
    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    vv[t] = v;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = u.m0u + t * 4 + rsel;
    if (col_ok && m < d.M) {
      if (d.out) *(float4*)(d.out + (size_t)m * d.ldo + col) = vv[t];
      if (d.out_sp) store_sp4((u16*)d.out_sp, (size_t)m, d.ldp, col, vv[t].x, vv[t].y, vv[t].z, vv[t].w);
    }
  }
}

// ---- MVD_EPI_QKV: every 32-column block of the tile lies inside one of q / k / v (heads * dhead is a multiple of 32: checked on the host).
//      q / k blocks: lane = (row member lane >> 3, 16-byte chunk lane & 7), two rounds of 8 rows; V^T blocks: lane = (column lane & 31,
//      row group lane >> 5), each lane 4 consecutive tokens of one channel per round, two rounds.
template <bool LNF>
__device__ __forceinline__ void pt_epi_qkv(const GemmParams& p, const PtUnit& u, int lane, float2 lnrow, unsigned ecnt_addr) {
  const mvd_gemm_desc& d = p.d;
  const int C = d.heads * d.dhead;
  const float scale = gemm_acc_scale(d);
  float4 x[4][2];
  int which[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int nb = u.n0 + b * 32;
    which[b] = nb < d.N ? nb / C : 3;                       // 3: block past N
    if (which[b] < 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) x[b][s] = st_ld4(u.st, u.rbase + s * 8 + (lane >> 3), b * 8 + (lane & 7));
    } else {
      const int col = lane & 31;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row = u.rbase + (s * 2 + (lane >> 5)) * 4;
        float t4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          t4[i] = *(const float*)(u.st + (row + i) * 512 + (((b * 8 + (col >> 2)) ^ ((row + i) & 15)) << 4) + (col & 3) * 4);
        x[b][s] = make_float4(t4[0], t4[1], t4[2], t4[3]);
      }
    }
  }
  lds_add1(ecnt_addr);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int nb = u.n0 + b * 32;
    if (which[b] == 3) continue;
    if (which[b] < 2) {
      const int dq = mvd_attn_dpad(d.dhead);
      u16* ph = (u16*)(which[b] == 0 ? d.q_hi : d.k_hi);
      u16* pl = (u16*)(which[b] == 0 ? d.q_lo : d.k_lo);
      const int n = nb + (lane & 7) * 4;
      const int cc = n - which[b] * C;
      const int head = cc / d.dhead, dd = cc - head * d.dhead;
      const float qs = which[b] == 0 ? d.qscale : 1.0f;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), cs = bb;
      if (d.bias) bb = ld4(d.bias + n);
      if (LNF) cs = ld4(d.ln_colsum + n);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int r = s * 8 + (lane >> 3);
        float4 v = x[b][s];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (LNF) {
          const float mean = __shfl(lnrow.x, r, 64), rstd = __shfl(lnrow.y, r, 64);
          v.x = (v.x - mean * cs.x) * rstd; v.y = (v.y - mean * cs.y) * rstd; v.z = (v.z - mean * cs.z) * rstd; v.w = (v.w - mean * cs.w) * rstd;
        }
        v = make_float4((v.x + bb.x) * qs, (v.y + bb.y) * qs, (v.z + bb.z) * qs, (v.w + bb.w) * qs);
        const int m = u.m0u + r;
        if (m < d.M) {
          const int bt = m / d.L, tok = m - bt * d.L;
          const size_t idx = ((size_t)(bt * d.heads + head) * d.Lpad + tok) * dq + dd;
          store_planes4(ph, pl, idx, v.x, v.y, v.z, v.w);
        }
      }
    } else {
      const int dv = (d.dhead + 15) & ~15;
      const int col = lane & 31;
      const int cc = nb + col - 2 * C;
      const int head = cc / d.dhead, dd = cc - head * d.dhead;
      const float bvv = d.bias ? d.bias[nb + col] : 0.f;
      const float csv = LNF ? d.ln_colsum[nb + col] : 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int r = (s * 2 + (lane >> 5)) * 4;
        float t4[4] = {x[b][s].x, x[b][s].y, x[b][s].z, x[b][s].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          t4[i] = t4[i] * scale;
          if (LNF) {
            const float mean = __shfl(lnrow.x, r + i, 64), rstd = __shfl(lnrow.y, r + i, 64);
            t4[i] = (t4[i] - mean * csv) * rstd;
          }
          t4[i] += bvv;
        }
        const int m = u.m0u + r;
        if (m < d.M) {
          const int bt = m / d.L, tok = m - bt * d.L;
          const size_t idx = ((size_t)(bt * d.heads + head) * dv + dd) * d.Lpad + tok;
          store_planes4((u16*)d.vt_hi, (u16*)d.vt_lo, idx, t4[0], t4[1], t4[2], t4[3]);
        }
      }
    }
  }
}

__device__ __forceinline__ void pt_epilogue(const GemmParams& p, unsigned char* smem, unsigned lds0, int i0, int i1, int e, int lane_in) {
  const mvd_gemm_desc& d = p.d;
  const unsigned fl = lds0 + OFF_FLAGS, abort_addr = fl + F_ABORT * 4;
  float* const ecol = (float*)(smem + OFF_ECOL) + e * 256;
  const int nitems = i1 - i0;
  int cur = -1;
  unsigned qafter = 0;                                      // k-tiles of items 0 .. cur
  PtItem w = {};
  const bool lnf = d.ln_stats != nullptr;
  [[maybe_unused]] long long st_ln = 0, st_wait = 0, st_epi = 0, st_units = 0;
  [[maybe_unused]] const long long st_t0 = PT_NOW();
  for (;;) {
    // (the lane index is made opaque per unit: otherwise the compiler hoists every per-lane address of every epilogue variant out of this
    //  loop and spills them)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const unsigned g = lds_ticket(fl + F_TICKET * 4);
    const int n = (int)(g >> 3), un = (int)(g & 7);
    if (n >= nitems) {
      if (e == 0) PT_DUMP(d, 2, PT_NOW() - st_t0, st_ln, st_wait, st_epi, st_units);
      return;
    }
    while (cur < n) {
      ++cur;
      w = pt_item(p, i0 + cur);
      qafter += w.nkt;
    }
    const int half = un >> 2;
    const unsigned qs = half == 0 ? qafter - 2 : qafter - 1;
    const int sslot = qs & 3;
    PtUnit u;
    u.st = smem + sslot * STAGE;
    u.rbase = (un & 3) * 16;
    u.m0u = w.m0 + un * 16;
    u.n0 = w.n0;
    u.z = w.z;
    float2 lnrow = make_float2(0.f, 1.f);
    PT_T(tl);
    if (lnf && lane < 16 && u.m0u + lane < d.M) lnrow = pt_ln_row(d, u.m0u + lane);     // (before the wait: its round trips hide behind the tile's k-loop)
    PT_ACC(st_ln, tl);
    PT_T(tw);
    if (!wait_ge(fl + (F_STAG + half) * 4, 4u * (unsigned)(n + 1), abort_addr)) return;
    PT_ACC(st_wait, tw);
    PT_T(te);
    const unsigned ecnt_addr = fl + (F_ECNT + sslot) * 4;
    if (MVD_PT_VARIANT & 8) {
      lds_add1(ecnt_addr);
    } else if (p.splits > 1) {
      pt_epi_slab(p, u, lane, ecnt_addr);
    } else if (d.epi == MVD_EPI_GEGLU) {
      if (lnf) pt_epi_geglu<true>(p, u, lane, lnrow, ecnt_addr);
      else pt_epi_geglu<false>(p, u, lane, lnrow, ecnt_addr);
    } else if (d.epi == MVD_EPI_QKV) {
      if (lnf) pt_epi_qkv<true>(p, u, lane, lnrow, ecnt_addr);
      else pt_epi_qkv<false>(p, u, lane, lnrow, ecnt_addr);
    } else if (d.act == MVD_ACT_NONE) {
      if (d.res) {
        if (d.bias_b) pt_epi_store<MVD_ACT_NONE, true, true>(p, u, lane, ecol, ecnt_addr);
        else pt_epi_store<MVD_ACT_NONE, true, false>(p, u, lane, ecol, ecnt_addr);
      } else {
        if (d.bias_b) pt_epi_store<MVD_ACT_NONE, false, true>(p, u, lane, ecol, ecnt_addr);
        else pt_epi_store<MVD_ACT_NONE, false, false>(p, u, lane, ecol, ecnt_addr);
      }
    } else if (d.act == MVD_ACT_SILU) {
      pt_epi_store<MVD_ACT_SILU, true, true>(p, u, lane, ecol, ecnt_addr);
    } else if (d.act == MVD_ACT_GELU) {
      pt_epi_store<MVD_ACT_GELU, true, true>(p, u, lane, ecol, ecnt_addr);
    } else {
      pt_epi_store<MVD_ACT_QUICKGELU, true, true>(p, u, lane, ecol, ecnt_addr);
    }
#ifdef MVD_PT_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PT_ACC(st_epi, te);
    ++st_units;
#endif
  }
}

// ============================================================================================== kernel
template <int NS, int AMODE>
__global__ __launch_bounds__(PT_THREADS) void gemm_pt_kernel(GemmParams p) {
  gemm_note_progress(p.d);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[PT_SMEM];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < F_COUNT) ((unsigned*)(smem + OFF_FLAGS))[tid] = 0;
  __syncthreads();                                           // the only workgroup barrier
  int i0, i1;
  pt_my_items(p, i0, i1);
  if (i0 >= i1) return;
#ifdef MVD_PT_STAMP
  if (blockIdx.x == 0 && lane == 0) ((long long*)p.d.workspace)[48 + wave] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
#endif
  if (wave < NCW) pt_consumer<NS>(p, smem, lds0, i0, i1, wave, lane);
  else if (wave < NCW + NLW) pt_loader<AMODE>(p, smem, lds0, i0, i1, wave - NCW, lane);
  else pt_epilogue(p, smem, lds0, i0, i1, wave - NCW - NLW, lane);
}

}  // namespace

int mvd_gemm_pt_min_ktiles() { return 2; }

bool mvd_gemm_pt_supported(const mvd_gemm_desc& d) {
  if (d.K < 64) return false;                                            // a tile's last TWO k-tiles lend their ring slots to the staging tile
  if (d.epi == MVD_EPI_STORE && (d.n_store & 3)) return false;           // (the ragged n_store edge stays with gemm_kernel)
  if (d.epi == MVD_EPI_QKV && ((d.heads * d.dhead) & 31)) return false;  // a 32-column block must lie inside one of q / k / v
  if (d.gn_stats && (d.gn_hw & 15)) return false;
  if (d.a_mode == MVD_A_DENSE && (size_t)d.M * d.lda * 4 + 4096 >= ((size_t)1 << 32)) return false;   // 32-bit lane offsets from a scalar base
  if (d.b_mode == MVD_B_PLANES && (size_t)d.N * d.ldb * 4 + 4096 >= ((size_t)1 << 32)) return false;
  if (d.b_mode == MVD_B_PACKED && (size_t)d.N * 128 + 4096 >= ((size_t)1 << 32)) return false;
  return true;
}

void mvd_gemm_pt_launch(const GemmParams& p, hipStream_t s) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;
  }
  const long T = (long)p.tiles_m * p.tiles_n * p.splits;
  const int G = (int)(T < cus ? T : cus);
  dim3 grid(G), block(PT_THREADS);
  const bool conv = p.d.a_mode == MVD_A_CONV3X3;
  const int ns = p.d.prec;
  if (!conv && ns == 4) hipLaunchKernelGGL((gemm_pt_kernel<4, MVD_A_DENSE>), grid, block, 0, s, p);
  if (!conv && ns == 3) hipLaunchKernelGGL((gemm_pt_kernel<3, MVD_A_DENSE>), grid, block, 0, s, p);
  if (!conv && ns == 1) hipLaunchKernelGGL((gemm_pt_kernel<1, MVD_A_DENSE>), grid, block, 0, s, p);
  if (conv && ns == 4) hipLaunchKernelGGL((gemm_pt_kernel<4, MVD_A_CONV3X3>), grid, block, 0, s, p);
  if (conv && ns == 3) hipLaunchKernelGGL((gemm_pt_kernel<3, MVD_A_CONV3X3>), grid, block, 0, s, p);
  if (conv && ns == 1) hipLaunchKernelGGL((gemm_pt_kernel<1, MVD_A_CONV3X3>), grid, block, 0, s, p);
}
