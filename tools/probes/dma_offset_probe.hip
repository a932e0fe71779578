// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
//   hipcc --offload-arch=gfx950 -O3 -o dma_offset_probe.bin tools/probes/dma_offset_probe.hip && ./dma_offset_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k(const unsigned* __restrict__ src, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned lds[4096];     // 16 KiB
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned voff = lane * 16;
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + 2048;      // M0 = byte 2048
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
      "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
      : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
  __syncthreads();
  for (int i = lane; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<unsigned> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = i;        // word i of the source holds i
  unsigned *src, *out;
  CHECK(hipMalloc(&src, 32768));
  CHECK(hipMalloc(&out, 16384));
  CHECK(hipMemcpy(src, h.data(), 32768, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned> o(4096);
  CHECK(hipMemcpy(o.data(), out, 16384, hipMemcpyDeviceToHost));
  // report runs of non-sentinel words: [lds word range] <- source word of the first
  int i = 0;
  while (i < 4096) {
    if (o[i] == 0xdeadbeefu) { ++i; continue; }
    int j = i;
    while (j + 1 < 4096 && o[j + 1] == o[j] + 1) ++j;
    printf("LDS bytes [%5d, %5d) <- source bytes [%5u, %5u)\n", i * 4, (j + 1) * 4, o[i] * 4, (o[j] + 1) * 4);
    i = j + 1;
  }
  printf("(M0 = 2048; loads with offset 0, 1024, 3072)\n");
  return 0;
}
