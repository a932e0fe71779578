// MFMA issue-rate probe (gfx950): back-to-back v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 on NACC independent accumulators,
// 1 or 2 wavefronts per SIMD, with and without a ds_read_b128 every few MFMAs.  Prints TFLOP/s per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

// 40 MFMAs per iteration on 10 accumulators (the 128x80 consumer tile: 2 A x 5 B fragments, 4 products); READS ds_read_b128 per
// iteration fill the fragment set of the NEXT iteration (register double buffering, as in the real k-loops): no MFMA waits for a read
// issued in the same iteration.
template <int NACC, int READS, int SYNC = 0>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  h8 fr[2][14];
  for (int j = 0; j < 14; ++j) fr[0][j] = fr[1][j] = (h8){1, 2, 3, 4, 5, 6, 7, 8};
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = 0.f;
  __syncthreads();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      if (SYNC == 1 || SYNC == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the k-loops' per-k-tile rendezvous
      if (SYNC == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < READS; ++j) {
        if (SYNC >= 3 && j < 4) {      // the GEMM's swizzled A-fragment pattern (row = lane & 15 of a 16-row block, chunk (lane >> 4 [+4]) ^ ((row >> 1) & 7))
          const int frow = lane & 15, fsw = (frow >> 1) & 7, fbase = (frow >> 3) * 1024 + (frow & 7) * 128;
          const int off = fbase + ((((j & 1) * 4 + (lane >> 4)) ^ fsw) * 16) + (j >> 1) * 2048;
          fr[par ^ 1][j] = *(const h8*)(lds + ((off + ((it + par) & 3) * 16384) & 65535));
        } else {
          fr[par ^ 1][j] = *(const h8*)(lds + ((lane * 16 + ((it + par) & 3) * 16384 + j * 1024) & 65535));
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[par][(i / 5) * 2 + (r >> 1)], fr[par][4 + (i % 5) * 2 + (r & 1)], acc[i], 0, 0, 0);
      }
      if constexpr (READS > 0) {
        constexpr int NM = 40;
        // one DS read after every (40 / READS) MFMAs
#pragma unroll
        for (int g = 0; g < READS; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, NM / READS, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
double timeit(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int t = 0; t < 3; ++t) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e-3;
}

int main() {
  float* out;
  hipMalloc(&out, 4 << 20);
  const int iters = 4000, blocks = 256;
  for (int threads : {256, 512}) {
    double t;
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 0>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD            : %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 5>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD +  5 ds_read_b128 per 40 MFMA (next-iteration fragments): %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 10>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD + 10 ds_read_b128 per 40 MFMA: %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 14>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD + 14 ds_read_b128 per 40 MFMA: %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 14, 1>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD + 14 ds_read_b128 + lgkmcnt(0) + s_barrier per 40 MFMA: %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 14, 2>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD + 14 ds_read_b128 + lgkmcnt(0) (no barrier) per 40 MFMA: %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 14, 3>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD + 14 ds_read_b128 (4 in the swizzled A pattern) + lgkmcnt(0) + s_barrier per 40 MFMA: %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k16<10, 0, 1>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("16x16x32  acc10  %d waves/SIMD + s_barrier per 40 MFMA (no reads): %7.1f TFLOP/s\n", threads / 256, 2.0 * 16 * 16 * 32 * 10 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k32<4>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("32x32x16  acc4   %d waves/SIMD            : %7.1f TFLOP/s\n", threads / 256, 2.0 * 32 * 32 * 16 * 4 * 4 * iters * (threads / 64) * blocks / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL((k32<8>), dim3(blocks), dim3(threads), 0, 0, out, iters); });
    printf("32x32x16  acc8   %d waves/SIMD            : %7.1f TFLOP/s\n", threads / 256, 2.0 * 32 * 32 * 16 * 8 * 4 * iters * (threads / 64) * blocks / t / 1e12);
  }
  return 0;
}
