// Shared device/host helpers for the mvd_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
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

__device__ __forceinline__ void split_bf16(float x, u16& hi, u16& lo) {
  // compiler-native conversions (v_cvt_pk_bf16_f32 on gfx950, round-to-nearest-even)
  const __bf16 h = (__bf16)x;
  const __bf16 l = (__bf16)(x - (float)h);
  hi = __builtin_bit_cast(u16, h);
  lo = __builtin_bit_cast(u16, l);
}

// four consecutive elements -> 8 bytes in each plane (idx must be a multiple of 4)
__device__ __forceinline__ void store_planes4(u16* __restrict__ hi, u16* __restrict__ lo, size_t idx, float a, float b,
                                              float c, float d) {
  u16 h[4], l[4];
  split_bf16(a, h[0], l[0]);
  split_bf16(b, h[1], l[1]);
  split_bf16(c, h[2], l[2]);
  split_bf16(d, h[3], l[3]);
  *(uint2*)(hi + idx) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
  *(uint2*)(lo + idx) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}
__device__ __forceinline__ void store_planes1(u16* __restrict__ hi, u16* __restrict__ lo, size_t idx, float a) {
  u16 h, l;
  split_bf16(a, h, l);
  hi[idx] = h;
  lo[idx] = l;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

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

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
