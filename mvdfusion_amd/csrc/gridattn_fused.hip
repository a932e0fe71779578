// Fused cross-view aggregation of GridAttn (mvdfusion/view_attn_efficient2.py:269-410): ONE launch per step does
//   G1-G3  depth sample -> unproject -> reproject into the V reference views and the input view -> bilinear gather ->
//          Plucker / harmonic embeddings          (the 723-wide token never leaves registers)
//   G4     Linear 723->256 + GELU, 3 x DiTBlock over the V views (adaLN-modulated LN, qkv, 8-head attention over V, proj,
//          MLP 256->512->256, gated residuals), weight_layer softmax over V, weighted sum
// and writes the pooled (Nseq, 256) rows as split planes for the last Linear 256->768 (a plain mvd_gemm).
//
// Mapping.  Token rows are ordered ((query view, pixel, depth sample), reference view slot): Vp consecutive rows = one 3-D point =
// one attention sequence, Vp = the next power of two >= V (so a sequence never straddles a wavefront's 16 rows).  Slots vr >= V are
// PADDING: they run the arithmetic on a copy of the last reference view (finite values), are masked out as attention KEYS and in the
// softmax-over-V pooling, and are never stored -- the reference's view counts 15 / 7 / 5 / 3 (configs/mvd_gso.yaml:97, mvd_train.yaml:90,97)
// take this kernel at 16 / 8 / 8 / 4 slots.  A wavefront owns 16 consecutive rows for the whole kernel; a workgroup is 4 wavefronts (64 rows).
// Activations never touch LDS or HBM: every GEMM is computed as  out^T = W x^T  (MFMA A operand = 16 weight rows, B operand =
// the wave's 16 activation rows), so a lane holds, for row (lane & 15), four consecutive output channels 16j + 4(lane>>4) + r
// of each 16-channel tile j -- exactly the register image of the NEXT GEMM's B fragment once the weight k-order inside
// every 32-block is permuted to (4g + r | 16 + 4g + r) at pack time.  LayerNorm / softmax / pooling statistics are a few
// cross-lane shuffles.  The value projection is computed un-transposed (A = activations) so that V^T is directly the A
// fragment of  O^T = V^T P^T  (16x16x16 MFMA), and S^T = K Q^T lands in the B-fragment layout of that MFMA.
//
// Weights (13 MB as fp16 hi + lo) are pre-packed on the host into ONE linear stream of 32 KiB slots in exactly the order
// the kernel consumes them (mvdfusion_amd/view_attn_efficient2.py: pack_fused_stream), already in the swizzled LDS image,
// so staging is a plain LDS-DMA copy (global_load_lds_dwordx4) into a 3-slot ring, two slots in flight ahead of the
// consumer, one `s_waitcnt vmcnt(8)` + barrier per slot (64 MFMAs per wave).  The small vectors (adaLN modulation of this
// step, biases, weight_layer) are DMA'd once into LDS, so the main loop issues no ordinary global load (those would drain
// the DMA queue at their vmcnt(0)).
#include <type_traits>

#include "gridattn_common.hpp"

namespace {

constexpr int G4_RING_SLOTS = 4;
constexpr int G4_SLOT_BYTES = 32768;                   // 16 micro-tiles (16 weight rows x 32 k, hi | lo = 2 KiB)
constexpr int G4_VEC_BLOCK = 3328;                     // floats per DiT block: mod 1536 | b_qkv 768 | b_proj 256 | b_fc1 512 | b_fc2 256 (13 KiB)
constexpr int G4_VEC_MISC = 3 * G4_VEC_BLOCK;          // (global layout) b_pre 256 | weight_layer w 256 | weight_layer b (1) ... | acc scales at +520
constexpr int G4_VEC_GRANULES = 44;                    // 44 KiB in global memory (11264 floats; 11008 used)
// LDS: a ring of FOUR weight slots (three in flight ahead of the consumer) + the misc vectors (4 KiB) + the vectors of TWO DiT blocks
// (13 KiB each, double-buffered: block b + 1 arrives while block b runs).  The kernel is bound by what one CU has in flight from the
// Infinity Cache: its 6.9 MB weight stream does not fit the XCD's 4 MiB L2, a slot takes ~2 us to arrive, and with a 3-slot ring (64 KiB in
// flight) a CU received 12 B/clk -- 2 700 cycles per slot against 768 cycles of MFMA work (r05_g4_time.log).  All three blocks' vectors
// resident (44 KiB) left no room for the fourth slot.
constexpr int G4_OFF_MISC = G4_RING_SLOTS * G4_SLOT_BYTES;
constexpr int G4_OFF_BLK = G4_OFF_MISC + 4096;
constexpr int G4_BLK_BYTES = G4_VEC_BLOCK * 4;         // 13 312 = 13 granules
constexpr int G4_SMEM = G4_OFF_BLK + 2 * G4_BLK_BYTES;
static_assert(G4_SMEM <= 160 * 1024 && G4_BLK_BYTES == 13 * 1024, "LDS budget");

// -DMVD_G4_STAMP (tools/probes/g4_stamp.sh): cycle accounting of workgroup 0 / wave 0 -- waiting for weight slots (DMA + barrier), inside
// the slots (fragment reads + MFMAs), everything else (the VALU phases) -- written to the buffer mvd_gridattn_fused_debug() names.
#ifdef MVD_G4_STAMP
#define G4_NOW() ((long long)__builtin_readcyclecounter())
#else
#define G4_NOW() 0ll
#endif

struct G4Params {
  long long* dbg;
  const float *x, *depth_noise, *steps;
  const int* iter;
  const float *grid_lin, *feat, *in_feat, *cams, *in_cam;
  const unsigned char* wstream;   // nslots x 32 KiB
  const float* vecs;              // G4_VEC_GRANULES * 256 floats
  u16* pooled_sp;                 // (Nseq, 256) split planes
  int V, Vp, lv, q0, Vq, S, D, nslots;      // Vp = 2^lv >= V: rows per 3-D point (reference views padded to a power of two)
  float depth_scale, depth_shift;
};

struct Frag {   // one MFMA operand fragment (8 elements) as hi + lo
  op16x8 hi, lo;
};

__device__ __forceinline__ Frag make_frag(const float (&v)[8]) {
  Frag f;
  union { op16x8 v; uint32_t e[4]; } H, L;
#pragma unroll
  for (int j = 0; j < 4; ++j) split_op16x2(v[2 * j], v[2 * j + 1], H.e[j], L.e[j]);      // (packed: 5 instructions per pair)
  f.hi = H.v;
  f.lo = L.v;
  return f;
}
__device__ __forceinline__ Frag make_frag(const f32x4& a, const f32x4& b) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return make_frag(v);
}

// acc += W^T-style product with all four partial products, same term order as gemm.hip (lo*lo, lo*hi, hi*lo, hi*hi with the
// activation as the first factor)
__device__ __forceinline__ void mma_lolo(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(w.lo, x.lo, acc, 0, 0, 0); }
__device__ __forceinline__ void mma_lohi(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(w.hi, x.lo, acc, 0, 0, 0); }
__device__ __forceinline__ void mma_hilo(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(w.lo, x.hi, acc, 0, 0, 0); }
__device__ __forceinline__ void mma_hihi(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(w.hi, x.hi, acc, 0, 0, 0); }
// un-transposed product (A = activations, B = weights): D[activation row][weight row]
__device__ __forceinline__ void mmu_lolo(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(x.lo, w.lo, acc, 0, 0, 0); }
__device__ __forceinline__ void mmu_lohi(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(x.lo, w.hi, acc, 0, 0, 0); }
__device__ __forceinline__ void mmu_hilo(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(x.hi, w.lo, acc, 0, 0, 0); }
__device__ __forceinline__ void mmu_hihi(f32x4& acc, const Frag& w, const Frag& x) { acc = MVD_MFMA_16x16x32(x.hi, w.hi, acc, 0, 0, 0); }

struct Taps {   // grid_sample(bilinear, border, align_corners=True): pixel offsets (in floats, 256 channels / pixel) and weights
  int o[4];
  float w[4];
};

__device__ __forceinline__ Taps make_taps(int S, float gx, float gy) {
  float ix = ((gx + 1.f) / 2.f) * (float)(S - 1);
  float iy = ((gy + 1.f) / 2.f) * (float)(S - 1);
  ix = fminf(fmaxf(ix, 0.f), (float)(S - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(S - 1));
  if (!(ix == ix)) ix = 0.f;
  if (!(iy == iy)) iy = 0.f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - x0f, wy1 = iy - y0f;
  const float wx0 = (x0f + 1.f) - ix, wy0 = (y0f + 1.f) - iy;
  Taps t;
  const int ys[4] = {y0, y0, y1, y1}, xs[4] = {x0, x1, x0, x1};
  const float ws[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool ok = ys[k] < S && xs[k] < S;
    t.o[k] = ok ? (ys[k] * S + xs[k]) * 256 : 0;
    t.w[k] = ok ? ws[k] : 0.f;      // bilinear4() skips such taps; a zero weight on an in-range pixel adds exactly 0
  }
  return t;
}

__device__ __forceinline__ f32x4 gather4(const float* __restrict__ fmap, const Taps& t, int ch) {
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 v = *(const float4*)(fmap + t.o[k] + ch);
    o[0] += v.x * t.w[k];
    o[1] += v.y * t.w[k];
    o[2] += v.z * t.w[k];
    o[3] += v.w * t.w[k];
  }
  return o;
}

// sin / cos with fp32-level accuracy for the embedding arguments (|a| <~ 30 rad), branch-free and small enough to be
// unrolled 56x per lane (ocml's sinf / cosf carry a large-argument path that drags scratch memory in): three-constant
// Cody-Waite reduction by pi/2 + the Cephes single-precision minimax polynomials on [-pi/4, pi/4] (~1 ulp).
__device__ __forceinline__ float sincos_sel(float a, bool want_cos) {
  const float k = rintf(a * 0.63661977236758134308f);
  float r = __builtin_fmaf(-k, 1.5703125f, a);
  r = __builtin_fmaf(-k, 4.837512969970703125e-4f, r);
  r = __builtin_fmaf(-k, 7.54978995489188216e-8f, r);
  const float z = r * r;
  const float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                                  __builtin_fmaf(-0.5f, z, 1.0f));
  const int q = ((int)k + (want_cos ? 1 : 0)) & 3;        // cos(a) = sin(a + pi/2)
  const float v = (q & 1) ? cp : sp;
  return (q & 2) ? -v : v;
}

// harmonic embedding element e of [sin(dim*7) | cos(dim*7) | x(dim)] without runtime-indexed arrays
struct Vec6 {   // named members (not an array): the selects below must stay selects, not become a runtime-indexed load
  float a, b, c, d, e, f;
};
__device__ __forceinline__ float pick6(float a, float b, float c, float d, float e, float f, int i) {   // by VALUE: SSA, never memory
  float r = a;
  r = i == 1 ? b : r;
  r = i == 2 ? c : r;
  r = i == 3 ? d : r;
  r = i == 4 ? e : r;
  r = i == 5 ? f : r;
  return r;
}
// token columns 512 + e: [ref plucker 90 | ref depth 15 | query plucker 90 | query depth 15 | 1 | zero pad]; each 105-block is
// [6-vector: sin 42 | cos 42 | x 6][scalar: sin 7 | cos 7 | x 1], harmonic index = dim * 7 + k, omega_k = 0.1 * 2^k
__device__ __forceinline__ float token_embedding(Vec6 rpl, float rdep, Vec6 qpl, float qdep, int e) {
  if (e >= 210) return e == 210 ? 1.0f : 0.0f;
  const bool query = e >= 105;
  const int i = query ? e - 105 : e;                     // index inside one 105-block
  const bool scalar = i >= 90;
  const int j = scalar ? i - 90 : i;                     // index inside the 90- or 15-value harmonic embedding
  const int n = scalar ? 7 : 42;                         // number of sin (= cos) entries
  const bool raw = j >= 2 * n;
  const int jj = j < n ? j : j - n;
  const int di = raw ? j - 2 * n : jj / 7;
  const int k = jj - (jj / 7) * 7;
  const float c6 = pick6(query ? qpl.a : rpl.a, query ? qpl.b : rpl.b, query ? qpl.c : rpl.c, query ? qpl.d : rpl.d,
                         query ? qpl.e : rpl.e, query ? qpl.f : rpl.f, di);
  const float comp = scalar ? (query ? qdep : rdep) : c6;
  const float sc = sincos_sel(comp * (0.1f * (float)(1 << k)), j >= n);
  return raw ? comp : sc;
}

template <int N>
__device__ __forceinline__ void g4_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory");
}

template <int NS>      // partial products per MAC: 4 = all of hi/lo x hi/lo, 3 = without lo*lo (the GEMMs' f16x3)
__global__ __launch_bounds__(256) void g4_fused_kernel(G4Params p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G4_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const float* const smisc = (const float*)(smem + G4_OFF_MISC);        // index = global index - G4_VEC_MISC
  const int V = p.V, Vp = p.Vp, lv = p.lv, S = p.S, D = p.D, SS = S * S;

  // ------------------------------------------------------------------ LDS-DMA engine
  // This wave copies granules [8 wave, 8 wave + 8) of every slot.  The copy is linear (source stride = destination stride = 1 KiB), so
  // the DMAs are issued in the scalar-base form with immediate offsets, which move source and destination alike
  // (tools/probes/dma_offset_probe.hip): M0 written twice per slot, 12 instructions for 8 KiB.  Through the builtin each DMA was a 64-bit
  // pointer add + s_mov m0 + load, 32 instructions per slot on a wave that issues one instruction per ~8 cycles and has only 48 MFMAs
  // (768 cycles) of work per slot.  Inline asm: hipcc does not count these; the counted waits are in g4_wait_barrier.
  const unsigned dma_voff = lane * 16;
  const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  auto dma8 = [&](const unsigned char* sbase, unsigned dst) {      // 8 KiB: sbase + lane * 16 + i KiB -> LDS dst + i KiB (lane-linear)
    unsigned keep;
    const unsigned v1 = dma_voff + 4096;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:2048\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:2048\n\tglobal_load_lds_dwordx4 %2, %3 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(dma_voff), "v"(v1), "s"(sbase), "s"(dst)
        : "memory", "scc");
  };
  const unsigned char* const wsrc = p.wstream + (size_t)wave * 8192;
  int s_issue = 0;                 // next slot to stage
  auto issue_slot = [&]() {
    if (s_issue < p.nslots) dma8(wsrc + (size_t)s_issue * G4_SLOT_BYTES, lds_ring + (s_issue % G4_RING_SLOTS) * G4_SLOT_BYTES + wave * 8192);
    ++s_issue;
  };
  // 1 - 4 consecutive granules (wave-uniform count)
  auto dma_run = [&](const unsigned char* sbase, unsigned dst, int cnt) {
    unsigned keep;
    if (cnt == 4)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(dma_voff), "s"(sbase), "s"(dst) : "memory");
    else if (cnt == 3)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(dma_voff), "s"(sbase), "s"(dst) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(dma_voff), "s"(sbase), "s"(dst) : "memory");
  };
  // the 13 granules of DiT block `blk`'s vectors -> LDS buffer blk & 1: wave 0 takes granules 0-3, waves 1-3 three each
  auto stage_block_vecs = [&](int blk) {
    const int g0 = wave == 0 ? 0 : 1 + 3 * wave, cnt = wave == 0 ? 4 : 3;
    dma_run((const unsigned char*)p.vecs + (size_t)blk * G4_BLK_BYTES + g0 * 1024, lds_ring + G4_OFF_BLK + (blk & 1) * G4_BLK_BYTES + g0 * 1024, cnt);
  };
  dma_run((const unsigned char*)p.vecs + (size_t)G4_VEC_MISC * 4 + wave * 1024, lds_ring + G4_OFF_MISC + wave * 1024, 1);      // misc: 4 granules
  stage_block_vecs(0);
  stage_block_vecs(1);
  bool stage_blk2 = false;         // set at the top of block 1: block 2's vectors replace block 0's behind the next barrier
  issue_slot();
  issue_slot();
  issue_slot();
  int s_cur = 0;                   // slot being consumed
  // acquire_wait: the LDS base of slot s_cur once it has landed for every wave; stage_ahead: DMA of the slot THREE ahead of the one being
  // consumed into the ring position of the slot consumed before it (every wave has passed its last read of that one: lgkmcnt(0) before the
  // barrier of acquire_wait).  Called between the first and second MFMA group of a slot, so that the
  // fragment reads of the slot go out first and the DMA instructions issue in the shadow of running MFMAs.
  [[maybe_unused]] long long st_wait = 0, st_slot = 0;
  [[maybe_unused]] const long long st_t0 = G4_NOW();
  auto acquire_wait = [&]() -> const unsigned char* {
    [[maybe_unused]] const long long tw = G4_NOW();
    // (counted: the DMAs of the slots issued AFTER slot s_cur may stay in flight -- two in the steady state; the vector DMAs in between only
    //  make the wait longer, never shorter: a wave's loads return in order)
    if (s_cur + 2 < p.nslots) g4_wait_barrier<16>();
    else if (s_cur + 1 < p.nslots) g4_wait_barrier<8>();
    else g4_wait_barrier<0>();
    if (stage_blk2) {              // every wave is past its last read of block 0's vectors (it has entered block 1)
      stage_block_vecs(2);
      stage_blk2 = false;
    }
    const unsigned char* base = smem + (s_cur % G4_RING_SLOTS) * G4_SLOT_BYTES;
    ++s_cur;
#ifdef MVD_G4_STAMP
    st_wait += G4_NOW() - tw;
#endif
    return base;
  };
  auto stage_ahead = [&]() { issue_slot(); };
  // fragment read offsets inside a 2 KiB micro-tile (same swizzle as gemm.hip): row R = lane & 15, 16-byte chunk g (hi) / 4 + g (lo)
  const int fsw = (r16 >> 1) & 7;
  const int fbase = (r16 >> 3) * 1024 + (r16 & 7) * 128;
  const int foff_hi = fbase + ((g) ^ fsw) * 16;
  const int foff_lo = fbase + ((4 + g) ^ fsw) * 16;
  auto read_w = [&](const unsigned char* slot, int mt) -> Frag {
    Frag f;
    f.hi = *(const op16x8*)(slot + mt * 2048 + foff_hi);
    f.lo = *(const op16x8*)(slot + mt * 2048 + foff_lo);
    return f;
  };
  auto read_group = [&](const unsigned char* slot, int gq, Frag (&w)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = read_w(slot, gq * 4 + q);
  };
  // One slot = 16 micro-tiles = 4 groups of 4 tiles (12 MFMAs per group: four independent accumulators per product term).  ONE wavefront
  // per SIMD runs this kernel (360+ registers), so nothing hides an LDS round trip it waits for: left to itself hipcc places each fragment
  // read right in front of the MFMA that needs it ("R w M R w M": 2 800 cycles per slot for 768 cycles of MFMA, 27 % of the pipe).  Here
  // the reads run two groups ahead of the MFMAs in three register sets, and scheduling barriers pin the phases
  //   R(g0) R(g1) R(g2) | M(g0) + DMA of the slot two ahead | R(g3) | M(g1) M(g2) | M(g3)
  // -- one exposed round trip per slot (its first group), the rest under 12 - 24 MFMAs; the counted lgkmcnt waits are the compiler's.
  // group_fn(integral_constant<gq>, w): the MFMAs of group gq (term-major over its four tiles, as before: same order per accumulator).
  auto run_slot = [&](auto&& group_fn) {
    using std::integral_constant;
    const unsigned char* sl = acquire_wait();
    [[maybe_unused]] const long long ts = G4_NOW();
    Frag wa[4], wb[4], wc[4];
    read_group(sl, 0, wa);
    read_group(sl, 1, wb);
    read_group(sl, 2, wc);
    __builtin_amdgcn_sched_barrier(0);
    group_fn(integral_constant<int, 0>{}, wa);
    stage_ahead();
    __builtin_amdgcn_sched_barrier(0);
    read_group(sl, 3, wa);
    __builtin_amdgcn_sched_barrier(0);
    group_fn(integral_constant<int, 1>{}, wb);
    group_fn(integral_constant<int, 2>{}, wc);
    __builtin_amdgcn_sched_barrier(0);
    group_fn(integral_constant<int, 3>{}, wa);
    __builtin_amdgcn_sched_barrier(0);
#ifdef MVD_G4_STAMP
    st_slot += G4_NOW() - ts;
#endif
  };

  // ------------------------------------------------------------------ G1-G3: this lane's row of the token matrix
  const size_t t_row = (size_t)blockIdx.x * 64 + wave * 16 + r16;
  const size_t pt = t_row >> lv;
  const int vslot = (int)(t_row & (size_t)(Vp - 1));
  const bool pad_row = vslot >= V;                       // padding slot: computed like the last real view, masked below
  const int vr = pad_row ? V - 1 : vslot;
  const int d = (int)(pt % D);
  const int pix = (int)((pt / D) % SS);
  const int b = p.q0 + (int)(pt / ((size_t)D * SS));
  // geometry of this lane's row: world point, Plucker coordinates, bilinear taps in the reference view and the input view
  Vec6 qpl, rpl;
  float rdep, depth;
  Taps tr, ti;
  {
    const int it = p.iter[0];
    const float sqrt_ac = p.steps[(size_t)it * MVD_STEP_STRIDE + 1];
    const float dstd = p.steps[(size_t)it * MVD_STEP_STRIDE + 2];
    const float dch = p.x[((size_t)b * 5 + 4) * SS + pix] / sqrt_ac;
    const float smp = dch + dstd * p.depth_noise[(((size_t)it * V + b) * D + d) * SS + pix];
    depth = fminf(fmaxf((smp + 1.0f) / 2.0f, 0.f), 1.f) * p.depth_scale + p.depth_shift;
    const Cam cb = load_cam(p.cams + (size_t)b * MVD_CAM_RECORD);
    const float ndx = p.grid_lin[pix % S], ndy = p.grid_lin[pix / S];
    float p1[3], p2[3], dir[3], org[3], X[3];
    unproject(cb, ndx, ndy, 1.f, p1);
    unproject(cb, ndx, ndy, 2.f, p2);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dir[j] = p2[j] - p1[j];
      org[j] = p1[j] - dir[j];
      X[j] = org[j] + depth * dir[j];
    }
    {
      const float nrm = fmaxf(sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]), 1e-12f);
      qpl.a = dir[0] / nrm;
      qpl.b = dir[1] / nrm;
      qpl.c = dir[2] / nrm;
      qpl.d = cb.C[1] * qpl.c - cb.C[2] * qpl.b;
      qpl.e = cb.C[2] * qpl.a - cb.C[0] * qpl.c;
      qpl.f = cb.C[0] * qpl.b - cb.C[1] * qpl.a;
    }
    const Cam cv = load_cam(p.cams + (size_t)vr * MVD_CAM_RECORD);
    {
      const float rd[3] = {X[0] - cv.C[0], X[1] - cv.C[1], X[2] - cv.C[2]};
      const float nr = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
      rdep = nr;
      const float nn = fmaxf(nr, 1e-12f);
      rpl.a = rd[0] / nn;
      rpl.b = rd[1] / nn;
      rpl.c = rd[2] / nn;
      rpl.d = cv.C[1] * rpl.c - cv.C[2] * rpl.b;
      rpl.e = cv.C[2] * rpl.a - cv.C[0] * rpl.c;
      rpl.f = cv.C[0] * rpl.b - cv.C[1] * rpl.a;
    }
    {
      float u, v;
      project(cv, X, u, v);
      tr = make_taps(S, -u, -v);
      const Cam ci = load_cam(p.in_cam);
      project(ci, X, u, v);
      ti = make_taps(S, -u, -v);
    }
  }
  const float* fref = p.feat + (size_t)vr * SS * 256;
  // ------------------------------------------------------------------ pre_layer: Linear(723 -> 256) + GELU
  f32x4 h[16];          // residual stream: tile j = channels 16j + 4g + r of row r16
  f32x4 acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // one slot = the 16 output tiles of one k-step, in groups of 4 tiles (four independent accumulators per MFMA burst)
  auto slot_16tiles = [&](const Frag& xf) {
    run_slot([&](auto gq_c, const Frag (&w)[4]) {
      constexpr int gq = decltype(gq_c)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if constexpr (NS == 4) mma_lolo(acc[gq * 4 + q], w[q], xf);
#pragma unroll
      for (int q = 0; q < 4; ++q) mma_lohi(acc[gq * 4 + q], w[q], xf);
#pragma unroll
      for (int q = 0; q < 4; ++q) mma_hilo(acc[gq * 4 + q], w[q], xf);
#pragma unroll
      for (int q = 0; q < 4; ++q) mma_hihi(acc[gq * 4 + q], w[q], xf);
    });
  };
  // The token fragments are produced in chunks of four k-steps right before they are consumed (all 23 at once would not fit
  // the register file); the ordinary loads of a chunk wait at vmcnt(0), i.e. once per chunk the two prefetched weight slots
  // simply finish landing first.
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    Frag tk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ks = (c4 & 1) * 4 + q;
      const float* fm = c4 < 2 ? fref : p.in_feat;
      const Taps& tp = c4 < 2 ? tr : ti;
      tk[q] = make_frag(gather4(fm, tp, 32 * ks + 4 * g), gather4(fm, tp, 32 * ks + 16 + 4 * g));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) slot_16tiles(tk[q]);
  }
  // embedding columns 512 .. 735: [ref plucker 90 | ref depth 15 | query plucker 90 | query depth 15 | 1 | 0 x 13]
#pragma unroll
  for (int ks = 0; ks < 7; ++ks) {
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      v[jj] = token_embedding(rpl, rdep, qpl, depth, 32 * ks + (jj < 4 ? 4 * g + jj : 16 + 4 * g + jj - 4));
    slot_16tiles(make_frag(v));
  }
  {
    const float sc = smisc[520];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 bb = *(const float4*)(smisc + 16 * j + 4 * g);
      float t0 = acc[j][0] * sc + bb.x, t1 = acc[j][1] * sc + bb.y, t2 = acc[j][2] * sc + bb.z, t3 = acc[j][3] * sc + bb.w;
      gelu_erf4(t0, t1, t2, t3);           // (packed polynomial: common.hpp)
      h[j] = (f32x4){t0, t1, t2, t3};
    }
  }

  // LayerNorm (eps 1e-6, no affine) + adaLN modulate of the wave's 16 rows -> the 8 B fragments of the next GEMM
  Frag xf[8];
  auto ln_modulate = [&](const float* shift, const float* scale) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += (h[j][0] + h[j][1]) + (h[j][2] + h[j][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s / 256.0f;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float a = h[j][0] - mean, b2 = h[j][1] - mean, c = h[j][2] - mean, d2 = h[j][3] - mean;
      q += (a * a + b2 * b2) + (c * c + d2 * d2);
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q / 256.0f + 1e-6f);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float v[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int j = 2 * ks + half;
        const float4 sc4 = *(const float4*)(scale + 16 * j + 4 * g), sh4 = *(const float4*)(shift + 16 * j + 4 * g);
        v[4 * half + 0] = (h[j][0] - mean) * rstd * (sc4.x + 1.f) + sh4.x;
        v[4 * half + 1] = (h[j][1] - mean) * rstd * (sc4.y + 1.f) + sh4.y;
        v[4 * half + 2] = (h[j][2] - mean) * rstd * (sc4.z + 1.f) + sh4.z;
        v[4 * half + 3] = (h[j][3] - mean) * rstd * (sc4.w + 1.f) + sh4.w;
      }
      xf[ks] = make_frag(v);
    }
  };
  // h += gate * (acc * acc_scale + bias), then clear acc
  auto gated_residual = [&](const float* gate, const float* bias, float sc) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 g4 = *(const float4*)(gate + 16 * j + 4 * g), b4 = *(const float4*)(bias + 16 * j + 4 * g);
      h[j][0] += g4.x * (acc[j][0] * sc + b4.x);
      h[j][1] += g4.y * (acc[j][1] * sc + b4.y);
      h[j][2] += g4.z * (acc[j][2] * sc + b4.z);
      h[j][3] += g4.w * (acc[j][3] * sc + b4.w);
      acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ------------------------------------------------------------------ 3 x DiTBlock (view_attn_efficient2.py:42-67)
  for (int blk = 0; blk < 3; ++blk) {
    const float* vb = (const float*)(smem + G4_OFF_BLK + (blk & 1) * G4_BLK_BYTES);
    if (blk == 1) stage_blk2 = true;
    const float* mod = vb;                 // shift_msa | scale_msa | gate_msa | shift_mlp | scale_mlp | gate_mlp
    const float sc_qkv = smisc[521 + blk * 4], sc_proj = smisc[522 + blk * 4];
    const float sc_fc1 = smisc[523 + blk * 4], sc_fc2 = smisc[524 + blk * 4];
    ln_modulate(mod, mod + 256);
    // ---- attention over the V views, head by head: qkv (3 slots) then this head's slice of proj (1 slot)
    for (int hd = 0; hd < 8; ++hd) {
      f32x4 qkv[6];      // q0 q1 k0 k1 (transposed layout: row r16, d = 16j + 4g + r) | v0 v1 (row 4g + r, d = 16j + r16)
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6) qkv[t6] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // micro-tile index within the head phase: mt = 16 sl3 + 4 gq + q -> k-step mt / 6, tile mt % 6
      auto qkv_slot = [&](auto sl3_c) {
        constexpr int sl3 = decltype(sl3_c)::value;
        run_slot([&](auto gq_c, const Frag (&w)[4]) {
          constexpr int gq = decltype(gq_c)::value;
#pragma unroll
          for (int term = 0; term < 4; ++term) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int mt = 16 * sl3 + 4 * gq + q;
              const int ks = mt / 6, t6 = mt - ks * 6;
              if (t6 < 4) {
                if (NS == 4 && term == 0) mma_lolo(qkv[t6], w[q], xf[ks]);
                if (term == 1) mma_lohi(qkv[t6], w[q], xf[ks]);
                if (term == 2) mma_hilo(qkv[t6], w[q], xf[ks]);
                if (term == 3) mma_hihi(qkv[t6], w[q], xf[ks]);
              } else {
                if (NS == 4 && term == 0) mmu_lolo(qkv[t6], w[q], xf[ks]);
                if (term == 1) mmu_lohi(qkv[t6], w[q], xf[ks]);
                if (term == 2) mmu_hilo(qkv[t6], w[q], xf[ks]);
                if (term == 3) mmu_hihi(qkv[t6], w[q], xf[ks]);
              }
            }
          }
        });
      };
      qkv_slot(std::integral_constant<int, 0>{});
      qkv_slot(std::integral_constant<int, 1>{});
      qkv_slot(std::integral_constant<int, 2>{});
      // bias (+ scale on q): timm Attention q = (x Wq^T + bq) * hd^-0.5
      const float* bq = vb + 1536 + 32 * hd;
      const float* bk = vb + 1536 + 256 + 32 * hd;
      const float* bv = vb + 1536 + 512 + 32 * hd;
      const float qs = 0.17677669529663687f;       // 32^-0.5
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 b4q = *(const float4*)(bq + 16 * j + 4 * g), b4k = *(const float4*)(bk + 16 * j + 4 * g);
        const float bvv = bv[16 * j + r16];
        const float bqa[4] = {b4q.x, b4q.y, b4q.z, b4q.w}, bka[4] = {b4k.x, b4k.y, b4k.z, b4k.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qkv[j][r] = (qkv[j][r] * sc_qkv + bqa[r]) * qs;
          qkv[2 + j][r] = qkv[2 + j][r] * sc_qkv + bka[r];
          qkv[4 + j][r] = qkv[4 + j][r] * sc_qkv + bvv;
        }
      }
      // S^T[key][query] = sum_d K[key][d] Q[query][d]   (one 16x16x32 MFMA per partial product)
      const Frag kf = make_frag(qkv[2], qkv[3]), qf = make_frag(qkv[0], qkv[1]);
      f32x4 st = {0.f, 0.f, 0.f, 0.f};
      if constexpr (NS == 4) st = MVD_MFMA_16x16x32(kf.lo, qf.lo, st, 0, 0, 0);
      st = MVD_MFMA_16x16x32(kf.lo, qf.hi, st, 0, 0, 0);
      st = MVD_MFMA_16x16x32(kf.hi, qf.lo, st, 0, 0, 0);
      st = MVD_MFMA_16x16x32(kf.hi, qf.hi, st, 0, 0, 0);
      // lane holds keys 4g + r of query r16; only keys of the query's own 3-D point (same group of V rows) take part
      float mx = -INFINITY;
      bool ok[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ok[r] = ((4 * g + r) >> lv) == (r16 >> lv) && ((4 * g + r) & (Vp - 1)) < V;
        if (ok[r]) mx = fmaxf(mx, st[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float den = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[r] = ok[r] ? expf(st[r] - mx) : 0.f;
        den += st[r];
      }
      den += __shfl_xor(den, 16, 64);
      den += __shfl_xor(den, 32, 64);
      // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]   (16x16x16 MFMA; A = V^T fragment = the un-transposed v tile)
      op4_t ph, pl;
      {
        union { op4_t v; uint32_t w[2]; } H, L;
        split_op16x2(st[0] / den, st[1] / den, H.w[0], L.w[0]);
        split_op16x2(st[2] / den, st[3] / den, H.w[1], L.w[1]);
        ph = H.v;
        pl = L.v;
      }
      f32x4 ot[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        union { op4_t v; uint32_t w[2]; } H, L;
        split_op16x2(qkv[4 + j][0], qkv[4 + j][1], H.w[0], L.w[0]);
        split_op16x2(qkv[4 + j][2], qkv[4 + j][3], H.w[1], L.w[1]);
        ot[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ot[j] = MVD_MFMA_16x16x16(L.v, pl, ot[j]);
        ot[j] = MVD_MFMA_16x16x16(L.v, ph, ot[j]);
        ot[j] = MVD_MFMA_16x16x16(H.v, pl, ot[j]);
        ot[j] = MVD_MFMA_16x16x16(H.v, ph, ot[j]);
      }
      // proj: this head's 32 input channels are k-step hd of W_proj
      const Frag of = make_frag(ot[0], ot[1]);
      slot_16tiles(of);
    }
    gated_residual(mod + 512, vb + 2304, sc_proj);
    // ---- MLP 256 -> 512 (GELU) -> 256 in chunks of 64 hidden channels: fc1 chunk (2 slots) then its two k-steps of fc2 (2 slots)
    ln_modulate(mod + 768, mod + 1024);
    for (int ch = 0; ch < 8; ++ch) {
      f32x4 f1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) f1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      auto fc1_slot = [&](auto sl2_c) {       // group = one k-step, the chunk's four 16-channel tiles
        constexpr int sl2 = decltype(sl2_c)::value;
        run_slot([&](auto gq_c, const Frag (&w)[4]) {
          constexpr int ks = 4 * sl2 + decltype(gq_c)::value;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if constexpr (NS == 4) mma_lolo(f1[q], w[q], xf[ks]);
#pragma unroll
          for (int q = 0; q < 4; ++q) mma_lohi(f1[q], w[q], xf[ks]);
#pragma unroll
          for (int q = 0; q < 4; ++q) mma_hilo(f1[q], w[q], xf[ks]);
#pragma unroll
          for (int q = 0; q < 4; ++q) mma_hihi(f1[q], w[q], xf[ks]);
        });
      };
      fc1_slot(std::integral_constant<int, 0>{});
      fc1_slot(std::integral_constant<int, 1>{});
      const float* b1 = vb + 2560 + 64 * ch;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *(const float4*)(b1 + 16 * q + 4 * g);
        float t0 = f1[q][0] * sc_fc1 + b4.x, t1 = f1[q][1] * sc_fc1 + b4.y, t2 = f1[q][2] * sc_fc1 + b4.z, t3 = f1[q][3] * sc_fc1 + b4.w;
        gelu_erf4(t0, t1, t2, t3);
        f1[q] = (f32x4){t0, t1, t2, t3};
      }
      slot_16tiles(make_frag(f1[0], f1[1]));
      slot_16tiles(make_frag(f1[2], f1[3]));
    }
    gated_residual(mod + 1280, vb + 3072, sc_fc2);
  }

  // ------------------------------------------------------------------ weight_layer + softmax over V + weighted sum (:83,396-397)
  {
    const float* wl = smisc + 256;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 w4 = *(const float4*)(wl + 16 * j + 4 * g);
      part += h[j][0] * w4.x + h[j][1] * w4.y + h[j][2] * w4.z + h[j][3] * w4.w;
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    const float lg = pad_row ? -INFINITY : part + smisc[512];
    // the Vp slots of a point are Vp consecutive lanes (r16): reduce over the low log2(Vp) lane bits (slot 0 is always real)
    float mx = lg;
    for (int o = 1; o < Vp; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = pad_row ? 0.f : expf(lg - mx);
    float den = e;
    for (int o = 1; o < Vp; o <<= 1) den += __shfl_xor(den, o, 64);
    const float pw = e / den;
    const size_t prow = t_row >> lv;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v0 = h[j][0] * pw, v1 = h[j][1] * pw, v2 = h[j][2] * pw, v3 = h[j][3] * pw;
      for (int o = 1; o < Vp; o <<= 1) {
        v0 += __shfl_xor(v0, o, 64);
        v1 += __shfl_xor(v1, o, 64);
        v2 += __shfl_xor(v2, o, 64);
        v3 += __shfl_xor(v3, o, 64);
      }
      if (vslot == 0) store_sp4(p.pooled_sp, prow, 256, 16 * j + 4 * g, v0, v1, v2, v3);
    }
  }
#ifdef MVD_G4_STAMP
  if (p.dbg && blockIdx.x == 0 && tid == 0) {
    p.dbg[0] = G4_NOW() - st_t0;
    p.dbg[1] = st_wait;
    p.dbg[2] = st_slot;
  }
#endif
}

}  // namespace

static long long* g_g4_dbg = nullptr;
#ifdef MVD_G4_STAMP
extern "C" void mvd_gridattn_fused_debug(long long* p) { g_g4_dbg = p; }
#endif
extern "C" int mvd_gridattn_fused_slots(void) { return 23 + 3 * 64; }
extern "C" size_t mvd_gridattn_fused_stream_bytes(void) { return (size_t)(23 + 3 * 64) * G4_SLOT_BYTES; }
extern "C" size_t mvd_gridattn_fused_vec_floats(void) { return (size_t)G4_VEC_GRANULES * 256; }

extern "C" int mvd_gridattn_fused(const float* x, const float* depth_noise, const float* steps, const int* iter,
                                  const float* grid_lin, const float* feat, const float* in_feat, const float* cams,
                                  const float* in_cam, const void* wstream, const float* vecs, void* pooled_sp, int V, int q0,
                                  int Vq, int S, int D, float depth_scale, float depth_shift, int prec, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && depth_noise && steps && iter && grid_lin && feat && in_feat && cams && in_cam && wstream && vecs && pooled_sp,
                "mvd_gridattn_fused: null pointer");
  MVD_CHECK_ARG(V >= 1 && V <= 16, "mvd_gridattn_fused: V=%d outside [1, 16] (use the unfused path)", V);
  int lv = 0;
  while ((1 << lv) < V) ++lv;
  const int Vp = 1 << lv;
  MVD_CHECK_ARG(q0 >= 0 && Vq > 0 && q0 + Vq <= V && S > 1 && D > 0, "mvd_gridattn_fused: bad shape");
  MVD_CHECK_ARG(prec == MVD_PREC_X3 || prec == MVD_PREC_X4, "mvd_gridattn_fused: prec %d (MVD_PREC_X3 or MVD_PREC_X4)", prec);
  MVD_CHECK_ARG(((uintptr_t)wstream & 15) == 0 && ((uintptr_t)vecs & 15) == 0 && ((uintptr_t)pooled_sp & 127) == 0,
                "mvd_gridattn_fused: wstream / vecs must be 16-byte, pooled_sp 128-byte aligned");
  const size_t T = (size_t)Vq * S * S * D * Vp;           // token rows incl. the padding slots
  MVD_CHECK_ARG(T % 64 == 0, "mvd_gridattn_fused: padded token count %zu must be a multiple of 64", T);
  G4Params p;
  p.dbg = g_g4_dbg;
  p.x = x; p.depth_noise = depth_noise; p.steps = steps; p.iter = iter; p.grid_lin = grid_lin; p.feat = feat;
  p.in_feat = in_feat; p.cams = cams; p.in_cam = in_cam; p.wstream = (const unsigned char*)wstream; p.vecs = vecs;
  p.pooled_sp = (u16*)pooled_sp; p.V = V; p.Vp = Vp; p.lv = lv; p.q0 = q0; p.Vq = Vq; p.S = S; p.D = D; p.nslots = 23 + 3 * 64;
  p.depth_scale = depth_scale; p.depth_shift = depth_shift;
  if (prec == MVD_PREC_X3) hipLaunchKernelGGL(g4_fused_kernel<3>, dim3((unsigned)(T / 64)), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(g4_fused_kernel<4>, dim3((unsigned)(T / 64)), dim3(256), 0, (hipStream_t)stream, p);
  MVD_CHECK_LAUNCH("mvd_gridattn_fused");
  return 0;
}
