// mvd_weight_prefetch (include/mvd_hip.h): one long-running kernel next to a graph-replayed step that reads every packed weight a few GEMM
// launches before its consumer, so that the consumer's k-loop meets Infinity-Cache hits (and warm translations) instead of a chain of HBM
// round trips.  Nothing is written; the loads' results are folded into a value that is stored only under a condition that never holds.
#include "common.hpp"
#include "../../include/mvd_hip.h"

namespace {

constexpr int PF_THREADS = 256;
constexpr int PF_UNROLL = 8;          // 16-byte loads in flight per thread: 32 KiB per workgroup and round

__global__ __launch_bounds__(PF_THREADS) void weight_prefetch_kernel(const mvd_prefetch_item* __restrict__ items, int n_items,
                                                                     const int* progress, int spin_limit, unsigned* sink) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  __shared__ int s_go;
  const int tid = threadIdx.x, nb = gridDim.x;
  unsigned fold = 0;
  for (int j = 0; j < n_items; ++j) {
    const mvd_prefetch_item it = items[j];
    if (tid == 0) {
      int go = -1, spins = 0;
      while (go < 0) {
        const int pr = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pr > it.consumer) go = 0;                 // too late: the consumer is already running
        else if (pr >= it.start_after) go = 1;
        else if (++spins > spin_limit) go = 2;        // the step stopped launching GEMMs: leave
        else __builtin_amdgcn_s_sleep(32);
      }
      s_go = go;
    }
    __syncthreads();
    const int go = s_go;
    __syncthreads();
    if (go == 2) break;
    if (go == 0) continue;
    // round r of this workgroup covers bytes [(r nb + blockIdx.x) CH, +CH), CH = PF_THREADS * PF_UNROLL * 16
    constexpr unsigned long long CH = (unsigned long long)PF_THREADS * PF_UNROLL * 16;
    const unsigned char* base = (const unsigned char*)it.ptr;
    for (unsigned long long off = (unsigned long long)blockIdx.x * CH; off < it.bytes; off += (unsigned long long)nb * CH) {
      u32x4 v[PF_UNROLL];
#pragma unroll
      for (int u = 0; u < PF_UNROLL; ++u) {
        const unsigned long long o = off + ((unsigned long long)u * PF_THREADS + tid) * 16;
        v[u] = o + 16 <= it.bytes ? *(const u32x4*)(base + o) : (u32x4){0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < PF_UNROLL; ++u) fold ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (sink != nullptr && fold == 0x9e3779b9u && n_items < 0) *sink = fold;      // (never true: keeps the loads alive)
}

}  // namespace

extern "C" int mvd_weight_prefetch(const mvd_prefetch_item* items, int n_items, const int* progress, int blocks, int spin_limit,
                                   mvd_stream_t stream) {
  MVD_CHECK_ARG(items != nullptr && progress != nullptr && n_items >= 0, "mvd_weight_prefetch: null table / progress counter");
  MVD_CHECK_ARG(blocks >= 1 && blocks <= 1024 && spin_limit > 0, "mvd_weight_prefetch: blocks=%d spin_limit=%d", blocks, spin_limit);
  if (n_items == 0) return 0;
  hipLaunchKernelGGL(weight_prefetch_kernel, dim3(blocks), dim3(PF_THREADS), 0, (hipStream_t)stream, items, n_items, progress, spin_limit,
                     (unsigned*)nullptr);
  MVD_CHECK_LAUNCH("mvd_weight_prefetch");
  return 0;
}
