// Shared between the GEMM translation units: gemm.hip (host entry mvd_gemm, split-K reduces, packing), gemm_plain_t0 ... t4.hip
// (gemm_kernel per block tile), gemm_ws.hip (role-split kernel), gemm_patch.hip (input-patch convolution).
#pragma once
#include "common.hpp"
#include "../../include/mvd_hip.h"

struct GemmParams {
  mvd_gemm_desc d;
  int nk;        // K / 32
  int nt16;      // packed N / 16
  int kt_per_split;
  int splits;
  int tiles_m, tiles_n;
  int m_fastest;
};

// acc_scale of the descriptor times its optional device scalar (mvd_gemm_desc.acc_scale_dev)
__device__ __forceinline__ float gemm_acc_scale(const mvd_gemm_desc& d) {
  return d.acc_scale_dev != nullptr ? d.acc_scale * *d.acc_scale_dev : d.acc_scale;
}

// Launchers of the tile-per-workgroup kernels, one per translation unit (grid = tiles x splits from GemmParams); false = the unit has no
// kernel for that tile / loop (include/mvd_hip.h: cfg).
bool mvd_gemm_launch_plain_t0(int loop, GemmParams& p, hipStream_t s);      // 64 x 64
bool mvd_gemm_launch_plain_t1(int loop, GemmParams& p, hipStream_t s);      // 128 x 128
bool mvd_gemm_launch_plain_t2(int loop, GemmParams& p, hipStream_t s);      // 128 x 80
bool mvd_gemm_launch_plain_t3(int loop, GemmParams& p, hipStream_t s);      // 64 x 80
bool mvd_gemm_launch_plain_t4(int loop, GemmParams& p, hipStream_t s);      // 128 x 160
bool mvd_gemm_launch_ws(int tile, GemmParams& p, hipStream_t s);            // gemm_ws_kernel: tiles 1, 2, 4
bool mvd_gemm_launch_patch(int tile, GemmParams& p, hipStream_t s, int patch_shares);      // conv_patch_kernel: tiles 1, 2, 4
