// Micro-benchmark: how fast can one CU pull L2-resident bytes into LDS, by LDS-DMA (global_load_lds_dwordx4) vs register
// staging (global_load_dwordx4 + ds_write_b128), for 4 / 8 waves per workgroup and 1 / 2 workgroups per CU?
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe tools/probes/dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each wave: `iters` rounds of U x 1 KiB pieces from its workgroup's source window (window bytes, L2-resident) into LDS
template <int MODE, int U>
__global__ void probe(const unsigned char* __restrict__ src, size_t window, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x % 64 * window;     // 64 distinct windows shared by the grid
  unsigned char* dst = lds + (wave * U % 64) * 1024;
  size_t off = (size_t)wave * U * 1024 + lane * 16;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off + u * 1024) % window),
                                         (__attribute__((address_space(3))) void*)(dst + (u % 8) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = *(const uint4*)(base + (off + u * 1024) % window);
#pragma unroll
      for (int u = 0; u < U; ++u) *(uint4*)(dst + (u % 8) * 1024 + lane * 16) = r[u];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    off += (size_t)nw * U * 1024;
  }
  acc = *(volatile unsigned*)(lds + lane * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int U>
int run(const char* name, const unsigned char* src, size_t window, unsigned* sink, int threads, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<MODE, U>), dim3(blocks), dim3(threads), 0, 0, src, window, 100, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<MODE, U>), dim3(blocks), dim3(threads), 0, 0, src, window, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * (threads / 64) * U * 1024.0 * iters;
  const double per_cu = bytes / (ms * 1e-3) / 256.0 / 1e9;
  printf("%-10s U=%d threads=%4d blocks=%4d  %8.1f GB/s per CU  (%5.1f B/clk @2.1GHz)  chip %6.2f TB/s\n", name, U, threads, blocks,
         per_cu * (256.0 / (blocks < 256 ? blocks : 256)), per_cu * (256.0 / (blocks < 256 ? blocks : 256)) / 2.1,
         bytes / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  const size_t window = 256 * 1024;
  unsigned char* src;
  unsigned* sink;
  CHECK(hipMalloc(&src, window * 64));
  CHECK(hipMemset(src, 1, window * 64));
  CHECK(hipMalloc(&sink, 64));
  for (int blocks : {256, 512}) {
    for (int threads : {256, 512}) {
      run<0, 4>("lds-dma", src, window, sink, threads, blocks);
      run<0, 8>("lds-dma", src, window, sink, threads, blocks);
      run<1, 4>("reg-stage", src, window, sink, threads, blocks);
      run<1, 8>("reg-stage", src, window, sink, threads, blocks);
    }
  }
  return 0;
}
