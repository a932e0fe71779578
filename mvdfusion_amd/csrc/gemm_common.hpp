// Shared between the GEMM translation units (gemm.hip: tile-per-workgroup kernels, split-K reduces, host entry; gemm_pt.hip: the
// persistent role-split kernel).
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

// gemm_pt.hip: persistent 128x128 kernel (mvd_gemm_desc.cfg loop 10).  pt_supported: whether it serves the problem; pt_launch enqueues it
// (the split-K reduce / GroupNorm kernels that may follow stay with mvd_gemm).
bool mvd_gemm_pt_supported(const mvd_gemm_desc& d);
int mvd_gemm_pt_min_ktiles();
void mvd_gemm_pt_launch(const GemmParams& p, hipStream_t s);
