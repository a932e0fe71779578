// gemm_kernel<64, 64, 2, 2, ...>: the loop variants of block tile 0 (include/mvd_hip.h: cfg), all three precisions, dense and 3x3-convolution A.
#include "gemm_plain.hpp"

bool mvd_gemm_launch_plain_t0(int loop, GemmParams& p, hipStream_t s) {
  switch (loop) {
    case 0: launch_cfg<64, 64, 2, 2, 2>(p, s); return true;
    case 1: launch_cfg<64, 64, 2, 2, 3>(p, s); return true;
    case 4: launch_cfg<64, 64, 2, 2, 6>(p, s); return true;
    case 5: launch_cfg<64, 64, 2, 2, 7>(p, s); return true;
  }
  return false;
}
