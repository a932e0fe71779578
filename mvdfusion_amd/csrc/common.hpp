// Shared device/host helpers for the mvd_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// MFMA operand element type.  Default: fp16 (11-bit significand; the 2-term split x ~= hi + lo then carries ~22 bits,
// i.e. fp32-class products).  -DMVD_OPERAND_BF16 builds the bf16 flavour (8-bit significand, ~16 bits split, scale-free
// exponent range).  Both run at the same MFMA rate on gfx950.
#ifdef MVD_OPERAND_BF16
typedef __bf16 op_t;
#define MVD_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MVD_OPERAND_FORMAT 0xbf16
#else
typedef _Float16 op_t;
#define MVD_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MVD_OPERAND_FORMAT 0xf16
#endif
typedef __attribute__((ext_vector_type(8))) op_t op16x8;   // 8 MFMA operand elements (fp16 or bf16: MVD_OPERAND_F16)
// the 16-deep MFMA (k = 16: lane group g = lane >> 4 holds k = 4g .. 4g + 3): attention over short sequences, 16-wide head-dim tails
#ifdef MVD_OPERAND_BF16
typedef __attribute__((ext_vector_type(4))) short op4_t;     // mfma_f32_16x16x16bf16_1k takes 4 x i16
#define MVD_MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)
#else
typedef __attribute__((ext_vector_type(4))) _Float16 op4_t;
#define MVD_MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0)
#endif
// head-dim padding of the attention operand planes: q / k rows and V^T row counts are rounded up to 16 (one MFMA k-step of 32 per full
// 32 channels plus a 16-deep tail step)
__host__ __device__ inline int mvd_attn_dpad(int dhead) { return (dhead + 15) & ~15; }
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

// ------------------------------------------------------------------------------------------------
// error state (C ABI: functions return 0 / negative, message via mvd_last_error())
// ------------------------------------------------------------------------------------------------
void mvd_set_error(const char* fmt, ...);

#define MVD_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      mvd_set_error(__VA_ARGS__);           \
      return -1;                            \
    }                                       \
  } while (0)

#define MVD_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t _e = hipGetLastError();                                           \
    if (_e != hipSuccess) {                                                      \
      mvd_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));       \
      return -2;                                                                 \
    }                                                                            \
  } while (0)

// ------------------------------------------------------------------------------------------------
// bf16 split:  x ~= hi + lo  with hi = rne_bf16(x), lo = rne_bf16(x - hi).  |x - hi - lo| <= 2^-18 |x|.
// Three MFMA products  hi*hi + hi*lo + lo*hi  then carry ~16 mantissa bits ("bf16x3").
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u16 f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  // NaN stays NaN; inf stays inf (adding the rounding bias to inf's zero mantissa is harmless)
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ void split_op16(float x, u16& hi, u16& lo) {
  // x ~= hi + lo in the operand type; compiler-native conversions (v_cvt_pk_*), round-to-nearest-even
  const op_t h = (op_t)x;
  const op_t l = (op_t)(x - (float)h);
  hi = __builtin_bit_cast(u16, h);
  lo = __builtin_bit_cast(u16, l);
}
// Two values at once, packed: hi2 = {hi(a) | hi(b) << 16}, lo2 likewise.  Written on 2-vectors so that hipcc emits the PACKED forms --
// v_cvt_pk_f16_f32 (both roundings to the operand type), v_pk_add_f32 (the residual): 5 instructions per pair against ~11 for two scalar
// splits + packing.  Same roundings, same results bit for bit; a wavefront issues one VALU instruction per ~7.5 cycles
// (profiles/r05_hw_facts_probe.log), and every epilogue, norm kernel and the fused GridAttn kernel split every value they hand to a GEMM.
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) op_t op16x2;
__device__ __forceinline__ void split_op16x2(float a, float b, uint32_t& hi2, uint32_t& lo2) {
  const f32x2 v = {a, b};
  const op16x2 h = __builtin_convertvector(v, op16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const op16x2 l = __builtin_convertvector(r, op16x2);
  hi2 = __builtin_bit_cast(uint32_t, h);
  lo2 = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ u16 to_op_bits(float x) { return __builtin_bit_cast(u16, (op_t)x); }

// ------------------------------------------------------------------------------------------------
// "Split planes" activation format (the A operand of mvd_gemm): a (rows, K) matrix, K % 32 == 0, stored per row as
// K/32 blocks of [32 x bf16 hi | 32 x bf16 lo] = 128 contiguous bytes per (row, 32-element k-block): exactly one cache
// line and exactly the unit the GEMM's LDS-DMA moves per row per k-tile.  Same bytes as fp32.
// Element (row, k): hi at  row*2*ld + (k>>5)*64 + (k&31)   (u16 units),  lo 32 u16 further.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t sp_index(size_t row, int ld, int k) { return row * 2 * (size_t)ld + (size_t)(k >> 5) * 64 + (k & 31); }

// four consecutive elements k..k+3 (k % 4 == 0) of one row
__device__ __forceinline__ void store_sp4(u16* __restrict__ base, size_t row, int ld, int k, float a, float b, float c, float d) {
  uint32_t h0, l0, h1, l1;
  split_op16x2(a, b, h0, l0);
  split_op16x2(c, d, h1, l1);
  u16* p = base + sp_index(row, ld, k);
  *(uint2*)p = make_uint2(h0, h1);
  *(uint2*)(p + 32) = make_uint2(l0, l1);
}
__device__ __forceinline__ void store_sp1(u16* __restrict__ base, size_t row, int ld, int k, float a) {
  u16 h, l;
  split_op16(a, h, l);
  u16* p = base + sp_index(row, ld, k);
  p[0] = h;
  p[32] = l;
}

// attention operand planes (separate hi / lo arrays): four consecutive elements -> 8 bytes in each plane
__device__ __forceinline__ void store_planes4(u16* __restrict__ hi, u16* __restrict__ lo, size_t idx, float a, float b,
                                              float c, float d) {
  uint32_t h0, l0, h1, l1;
  split_op16x2(a, b, h0, l0);
  split_op16x2(c, d, h1, l1);
  *(uint2*)(hi + idx) = make_uint2(h0, h1);
  *(uint2*)(lo + idx) = make_uint2(l0, l1);
}
__device__ __forceinline__ void store_planes1(u16* __restrict__ hi, u16* __restrict__ lo, size_t idx, float a) {
  u16 h, l;
  split_op16(a, h, l);
  hi[idx] = h;
  lo[idx] = l;
}

// erf without branches: erf(|x|) = 1 - exp(-t(|x|)), t = -ln erfc fitted by |x| * P8(|x|) on [0, 4] (erfc(4) < 2^-25: erf == 1 in fp32
// from there on).  Max absolute error 9.5e-8 over the whole line (the rounding of 1 - exp dominates), i.e. the level of an fp32
// rounding of erf itself -- which is all the exact GELU 0.5 x (1 + erf(x / sqrt 2)) needs, down to x -> 0.  15 VALU operations and
// one basic block (the library erff is two regimes behind a divergent branch: ~35 operations, and a scheduling boundary that keeps
// MFMAs of the surrounding GEMM phases from overlapping it -- csrc/gridattn_fused.hip evaluates 448 GELUs per lane).
__device__ __forceinline__ float erf_nobranch(float x) {
  const float a = fminf(fabsf(x), 4.0f);
  float p = -8.043882189667784e-06f;
  p = __builtin_fmaf(p, a, 0.00010602718248264864f);
  p = __builtin_fmaf(p, a, -0.0005879526142962277f);
  p = __builtin_fmaf(p, a, 0.0015767638105899096f);
  p = __builtin_fmaf(p, a, -5.878534648218192e-05f);
  p = __builtin_fmaf(p, a, -0.01921714097261429f);
  p = __builtin_fmaf(p, a, 0.10279920697212219f);
  p = __builtin_fmaf(p, a, 0.6366161108016968f);
  p = __builtin_fmaf(p, a, 1.1283793449401855f);
  const float e = __builtin_amdgcn_exp2f(p * a * -1.4426950408889634f);
  return copysignf(1.0f - e, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_nobranch(x * 0.70710678118654752440f)); }
// Two GELUs at once with the polynomial on PACKED fp32 (v_pk_fma_f32 / v_pk_mul_f32: one instruction per pair): the same operations in the
// same order as gelu_erf on each element => bit-identical results, ~2/3 of the VALU instructions.  (The erf-GELU epilogue of the GEGLU
// projections is VALU-issue bound: DESIGN.md section 6.)
typedef __attribute__((ext_vector_type(2))) float mvd_f32x2;
__device__ __forceinline__ void gelu_erf2(float& x0, float& x1) {
  const mvd_f32x2 x = {x0, x1};
  const mvd_f32x2 t = x * 0.70710678118654752440f;
  const mvd_f32x2 a = {fminf(fabsf(t.x), 4.0f), fminf(fabsf(t.y), 4.0f)};
  mvd_f32x2 p = {-8.043882189667784e-06f, -8.043882189667784e-06f};
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){0.00010602718248264864f, 0.00010602718248264864f});
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){-0.0005879526142962277f, -0.0005879526142962277f});
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){0.0015767638105899096f, 0.0015767638105899096f});
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){-5.878534648218192e-05f, -5.878534648218192e-05f});
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){-0.01921714097261429f, -0.01921714097261429f});
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){0.10279920697212219f, 0.10279920697212219f});
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){0.6366161108016968f, 0.6366161108016968f});
  p = __builtin_elementwise_fma(p, a, (mvd_f32x2){1.1283793449401855f, 1.1283793449401855f});
  const mvd_f32x2 arg = p * a * -1.4426950408889634f;
  const float e0 = __builtin_amdgcn_exp2f(arg.x), e1 = __builtin_amdgcn_exp2f(arg.y);
  const mvd_f32x2 erf = {copysignf(1.0f - e0, t.x), copysignf(1.0f - e1, t.y)};
  const mvd_f32x2 r = 0.5f * x * (1.0f + erf);
  x0 = r.x;
  x1 = r.y;
}
__device__ __forceinline__ void gelu_erf4(float& a, float& b, float& c, float& d) {
  gelu_erf2(a, b);
  gelu_erf2(c, d);
}
// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (~10 instructions: div_scale / rcp / 3 fma / div_fmas /
// div_fixup): the GroupNorm-apply kernels evaluate it for every element of 55 tensors per step.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// GroupNorm statistics handed from a PRODUCER kernel (GEMM epilogue, split-K reduce, concat) to mvd_groupnorm_from_stats: per
// (image, group) {sum, sum of squares} accumulated as 64-bit fixed point (value * 2^24) with integer atomics -- the order of the
// contributions does not matter, so the result is deterministic; every contribution is an fp32 partial sum formed in a fixed
// order, represented exactly down to 2^-24.  Capacity: |total| < 5.5e11.
#define MVD_GN_FIXED_SCALE 16777216.0f
__device__ __forceinline__ void gn_stats_add(long long* stats, int b, int g, int groups, float s, float q) {
  unsigned long long* p = (unsigned long long*)(stats + ((size_t)b * groups + g) * 2);
  atomicAdd(p, (unsigned long long)(long long)llrintf(s * MVD_GN_FIXED_SCALE));
  atomicAdd(p + 1, (unsigned long long)(long long)llrintf(q * MVD_GN_FIXED_SCALE));
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per DEVICE (function attributes may be per device: a model moved to another GPU of
// the same process must get it too -- ADVICE r04).  `done` is a per-kernel bitmask of device ordinals (ordinals >= 64 set it every time).
static inline hipError_t mvd_raise_dynamic_lds(const void* fn, int bytes, unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < 64 && ((*done >> dev) & 1ull)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && dev >= 0 && dev < 64) *done |= 1ull << dev;
  return e;
}
