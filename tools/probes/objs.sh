# Object lists of libmvd_hip.so (sourced by the probe build scripts; paths relative to mvdfusion_amd/csrc).
GEMM_OBJS="gemm.o gemm_plain_t0.o gemm_plain_t1.o gemm_plain_t2.o gemm_plain_t3.o gemm_plain_t4.o gemm_ws.o gemm_patch.o"
REST_OBJS="api.o prefetch.o norm.o attention.o elementwise.o gridattn.o backward.o"
