// GridAttn front end (mvdfusion/view_attn_efficient2.py:269-370, 413-437): z-embedding of the latents and the fused
// depth-sample -> unproject -> reproject -> bilinear gather -> Plucker/harmonic embedding kernel that writes the token
// matrix consumed by the aggregation transformer's first GEMM.
//
// One wavefront per 3-D query point (query view b, pixel, depth sample d); it loops over the V reference views and
// writes one coalesced 736-float row per view: lanes own 4 feature channels each (float4 gathers from the
// channels-last feature maps, which stay L2/MALL resident: (V+1) x S x S x 256 fp32 = 1 MB per view), and the 210
// sin/cos embedding values are spread over the lanes.
#include "gridattn_common.hpp"

namespace {

__global__ __launch_bounds__(256) void tokens_kernel(const float* __restrict__ x, const float* __restrict__ depth_noise,
                                                     const float* __restrict__ steps, const int* __restrict__ iter,
                                                     const float* __restrict__ grid_lin, const float* __restrict__ feat,
                                                     const float* __restrict__ in_feat, const float* __restrict__ cams,
                                                     const float* __restrict__ in_cam, u16* __restrict__ tok, int V, int q0, int Vq, int S, int D, float depth_scale, float depth_shift) {
  const int lane = threadIdx.x & 63;
  const int SS = S * S;
  const size_t npts = (size_t)Vq * SS * D;
  const size_t pt = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= npts) return;
  const int d = (int)(pt % D);
  const int pix = (int)((pt / D) % SS);
  const int b = q0 + (int)(pt / ((size_t)D * SS));  // global index of the query view
  const int it = iter[0];
  const float sqrt_ac = steps[(size_t)it * MVD_STEP_STRIDE + 1];
  const float dstd = steps[(size_t)it * MVD_STEP_STRIDE + 2];

  // ---- G1: depth sample and world point  (:419-432, ray_utils.py:175-202,367-369)
  const float dch = x[((size_t)b * 5 + 4) * SS + pix] / sqrt_ac;
  const float smp = dch + dstd * depth_noise[(((size_t)it * V + b) * D + d) * SS + pix];
  const float depth = fminf(fmaxf((smp + 1.0f) / 2.0f, 0.f), 1.f) * depth_scale + depth_shift;
  const Cam cb = load_cam(cams + (size_t)b * MVD_CAM_RECORD);
  const float ndx = grid_lin[pix % S], ndy = grid_lin[pix / S];
  float p1[3], p2[3], dir[3], org[3], X[3];
  unproject(cb, ndx, ndy, 1.f, p1);
  unproject(cb, ndx, ndy, 2.f, p2);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    dir[j] = p2[j] - p1[j];
    org[j] = p1[j] - dir[j];
    X[j] = org[j] + depth * dir[j];
  }
  // ---- query-side geometry (same for every reference view)  (:344-362)
  float qpl[6], qd[1] = {depth};
  {
    const float nrm = fmaxf(sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]), 1e-12f);
    qpl[0] = dir[0] / nrm;
    qpl[1] = dir[1] / nrm;
    qpl[2] = dir[2] / nrm;
    qpl[3] = cb.C[1] * qpl[2] - cb.C[2] * qpl[1];
    qpl[4] = cb.C[2] * qpl[0] - cb.C[0] * qpl[2];
    qpl[5] = cb.C[0] * qpl[1] - cb.C[1] * qpl[0];
  }
  float qe0, qe1 = 0.f;  // this lane's two query-embedding values (105 = 90 + 15)
  {
    const int e0 = lane, e1 = lane + 64;
    qe0 = e0 < 90 ? harmonic(qpl, 6, e0) : harmonic(qd, 1, e0 - 90);
    if (e1 < 105) qe1 = e1 < 90 ? harmonic(qpl, 6, e1) : harmonic(qd, 1, e1 - 90);
  }
  // ---- input-view gather (same for every reference view)  (:320-331)
  float4 fin;
  {
    const Cam ci = load_cam(in_cam);
    float u, v;
    project(ci, X, u, v);
    fin = bilinear4(in_feat, S, lane * 4, -u, -v);
  }
  // ---- per reference view
  for (int vr = 0; vr < V; ++vr) {
    const Cam cv = load_cam(cams + (size_t)vr * MVD_CAM_RECORD);
    float u, v;
    project(cv, X, u, v);
    const float4 fr = bilinear4(feat + (size_t)vr * SS * 256, S, lane * 4, -u, -v);
    float rd[3] = {X[0] - cv.C[0], X[1] - cv.C[1], X[2] - cv.C[2]};
    const float nr = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
    float rdep[1] = {nr};
    const float nn = fmaxf(nr, 1e-12f);
    float rpl[6];
    rpl[0] = rd[0] / nn;
    rpl[1] = rd[1] / nn;
    rpl[2] = rd[2] / nn;
    rpl[3] = cv.C[1] * rpl[2] - cv.C[2] * rpl[1];
    rpl[4] = cv.C[2] * rpl[0] - cv.C[0] * rpl[2];
    rpl[5] = cv.C[0] * rpl[1] - cv.C[1] * rpl[0];
    const size_t row = pt * V + vr;
    store_sp4(tok, row, MVD_TOKEN_LD, lane * 4, fr.x, fr.y, fr.z, fr.w);
    store_sp4(tok, row, MVD_TOKEN_LD, 256 + lane * 4, fin.x, fin.y, fin.z, fin.w);
    {
      const int e0 = lane, e1 = lane + 64;
      store_sp1(tok, row, MVD_TOKEN_LD, 512 + e0, e0 < 90 ? harmonic(rpl, 6, e0) : harmonic(rdep, 1, e0 - 90));
      if (e1 < 105) store_sp1(tok, row, MVD_TOKEN_LD, 512 + e1, e1 < 90 ? harmonic(rpl, 6, e1) : harmonic(rdep, 1, e1 - 90));
      store_sp1(tok, row, MVD_TOKEN_LD, 617 + e0, qe0);
      if (e1 < 105) store_sp1(tok, row, MVD_TOKEN_LD, 617 + e1, qe1);
    }
    if (lane < MVD_TOKEN_LD - 722) store_sp1(tok, row, MVD_TOKEN_LD, 722 + lane, lane == 0 ? 1.0f : 0.0f);  // mask = 1, zero pad
  }
}

// Linear(5 -> 256) + GELU per pixel; lat (N,5,S,S) NCHW -> feat (N,S,S,256) NHWC; one wave per pixel
__global__ __launch_bounds__(256) void zembed_kernel(const float* __restrict__ lat, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ feat, int N, int SS) {
  const int lane = threadIdx.x & 63;
  const size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= (size_t)N * SS) return;
  const int n = (int)(p / SS), pix = (int)(p % SS);
  float in[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) in[c] = lat[((size_t)n * 5 + c) * SS + pix];
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = lane * 4 + j;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) a += in[c] * w[ch * 5 + c];
    o[j] = gelu_erf(a + bias[ch]);
  }
  *(float4*)(feat + p * 256 + lane * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------ token kernel backward
// Gradient of the token matrix's two gathered blocks w.r.t. the feature maps (F.grid_sample backward w.r.t. its input;
// view_attn_efficient2.py:320-341): dtok (T, ldt) fp32 = dL/d tokens from the first GEMM's dgrad; columns [0, 256) were
// bilinear samples of feat[vr], [256, 512) of in_feat.  Scatter with the forward's own taps and weights into 64-bit fixed-point
// accumulators (value * scale, integer atomics: order independent => bit-reproducible), converted by the caller.
// The sampling positions depend only on data (noisy latents, cameras), never on parameters: no gradient flows there.
struct Taps4 {
  int idx[4];
  float w[4];
};
__device__ __forceinline__ Taps4 bilinear_taps(int S, float gx, float gy) {
  float ix = ((gx + 1.f) / 2.f) * (float)(S - 1);
  float iy = ((gy + 1.f) / 2.f) * (float)(S - 1);
  ix = fminf(fmaxf(ix, 0.f), (float)(S - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(S - 1));
  if (!(ix == ix)) ix = 0.f;
  if (!(iy == iy)) iy = 0.f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - x0f, wy1 = iy - y0f;
  const float wx0 = (x0f + 1.f) - ix, wy0 = (y0f + 1.f) - iy;
  Taps4 t;
  t.idx[0] = (y0 < S && x0 < S) ? y0 * S + x0 : -1;
  t.idx[1] = (y0 < S && x1 < S) ? y0 * S + x1 : -1;
  t.idx[2] = (y1 < S && x0 < S) ? y1 * S + x0 : -1;
  t.idx[3] = (y1 < S && x1 < S) ? y1 * S + x1 : -1;
  t.w[0] = wx0 * wy0;
  t.w[1] = wx1 * wy0;
  t.w[2] = wx0 * wy1;
  t.w[3] = wx1 * wy1;
  return t;
}
__device__ __forceinline__ void scatter4(long long* __restrict__ acc, const Taps4& t, int ch, float4 g, float scale) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (t.idx[k] < 0) continue;
    unsigned long long* p = (unsigned long long*)(acc + (size_t)t.idx[k] * 256 + ch);
    const float w = t.w[k] * scale;
    atomicAdd(p + 0, (unsigned long long)(long long)llrintf(g.x * w));
    atomicAdd(p + 1, (unsigned long long)(long long)llrintf(g.y * w));
    atomicAdd(p + 2, (unsigned long long)(long long)llrintf(g.z * w));
    atomicAdd(p + 3, (unsigned long long)(long long)llrintf(g.w * w));
  }
}

__global__ __launch_bounds__(256) void tokens_bwd_kernel(const float* __restrict__ x, const float* __restrict__ depth_noise,
                                                         const float* __restrict__ steps, const int* __restrict__ iter,
                                                         const float* __restrict__ grid_lin, const float* __restrict__ cams,
                                                         const float* __restrict__ in_cam, const float* __restrict__ dtok, int ldt,
                                                         long long* __restrict__ dfeat, long long* __restrict__ din_feat, float scale, int V,
                                                         int q0, int Vq, int S, int D, float depth_scale, float depth_shift) {
  const int lane = threadIdx.x & 63;
  const int SS = S * S;
  const size_t npts = (size_t)Vq * SS * D;
  const size_t pt = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= npts) return;
  const int d = (int)(pt % D);
  const int pix = (int)((pt / D) % SS);
  const int b = q0 + (int)(pt / ((size_t)D * SS));
  const int it = iter[0];
  const float sqrt_ac = steps[(size_t)it * MVD_STEP_STRIDE + 1];
  const float dstd = steps[(size_t)it * MVD_STEP_STRIDE + 2];
  // the forward's G1 geometry, verbatim (tokens_kernel)
  const float dch = x[((size_t)b * 5 + 4) * SS + pix] / sqrt_ac;
  const float smp = dch + dstd * depth_noise[(((size_t)it * V + b) * D + d) * SS + pix];
  const float depth = fminf(fmaxf((smp + 1.0f) / 2.0f, 0.f), 1.f) * depth_scale + depth_shift;
  const Cam cb = load_cam(cams + (size_t)b * MVD_CAM_RECORD);
  const float ndx = grid_lin[pix % S], ndy = grid_lin[pix / S];
  float p1[3], p2[3], X[3];
  unproject(cb, ndx, ndy, 1.f, p1);
  unproject(cb, ndx, ndy, 2.f, p2);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float dir = p2[j] - p1[j];
    X[j] = (p1[j] - dir) + depth * dir;
  }
  float4 gin = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int vr = 0; vr < V; ++vr) {
    const Cam cv = load_cam(cams + (size_t)vr * MVD_CAM_RECORD);
    float u, v;
    project(cv, X, u, v);
    const Taps4 t = bilinear_taps(S, -u, -v);
    const float* row = dtok + (pt * V + vr) * (size_t)ldt;
    const float4 g = *(const float4*)(row + lane * 4);
    scatter4(dfeat + (size_t)vr * SS * 256, t, lane * 4, g, scale);
    const float4 gi = *(const float4*)(row + 256 + lane * 4);     // the input-view block is the same sample in all V rows
    gin.x += gi.x; gin.y += gi.y; gin.z += gi.z; gin.w += gi.w;
  }
  {
    const Cam ci = load_cam(in_cam);
    float u, v;
    project(ci, X, u, v);
    scatter4(din_feat, bilinear_taps(S, -u, -v), lane * 4, gin, scale);
  }
}

}  // namespace

extern "C" int mvd_zembed(const float* lat, const float* w, const float* b, float* feat, int N, int S, mvd_stream_t stream) {
  MVD_CHECK_ARG(lat && w && b && feat && N > 0 && S > 0, "mvd_zembed: bad arguments");
  const size_t npix = (size_t)N * S * S;
  hipLaunchKernelGGL(zembed_kernel, dim3(cdiv(npix, 4)), dim3(256), 0, (hipStream_t)stream, lat, w, b, feat, N, S * S);
  MVD_CHECK_LAUNCH("mvd_zembed");
  return 0;
}

extern "C" int mvd_gridattn_tokens(const float* x, const float* depth_noise, const float* steps, const int* iter,
                                   const float* grid_lin, const float* feat, const float* in_feat, const float* cams,
                                   const float* in_cam, void* tokens_sp, int V, int q0, int Vq, int S, int D, float depth_scale,
                                   float depth_shift, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && depth_noise && steps && iter && grid_lin && feat && in_feat && cams && in_cam && tokens_sp,
                "mvd_gridattn_tokens: null pointer");
  MVD_CHECK_ARG(V > 0 && V <= 16 && S > 1 && D > 0, "mvd_gridattn_tokens: bad shape (V <= 16)");
  MVD_CHECK_ARG(q0 >= 0 && Vq > 0 && q0 + Vq <= V, "mvd_gridattn_tokens: bad query-view range [%d, %d) of %d", q0, q0 + Vq, V);
  const size_t npts = (size_t)Vq * S * S * D;
  hipLaunchKernelGGL(tokens_kernel, dim3(cdiv(npts, 4)), dim3(256), 0, (hipStream_t)stream, x, depth_noise, steps, iter,
                     grid_lin, feat, in_feat, cams, in_cam, (u16*)tokens_sp, V, q0, Vq, S, D, depth_scale, depth_shift);
  MVD_CHECK_LAUNCH("mvd_gridattn_tokens");
  return 0;
}

extern "C" int mvd_gridattn_tokens_backward(const float* x, const float* depth_noise, const float* steps, const int* iter,
                                            const float* grid_lin, const float* cams, const float* in_cam, const float* dtok, int ldt,
                                            long long* dfeat_acc, long long* din_feat_acc, float scale, int V, int q0, int Vq, int S, int D,
                                            float depth_scale, float depth_shift, mvd_stream_t stream) {
  MVD_CHECK_ARG(x && depth_noise && steps && iter && grid_lin && cams && in_cam && dtok && dfeat_acc && din_feat_acc,
                "mvd_gridattn_tokens_backward: null pointer");
  MVD_CHECK_ARG(V > 0 && V <= 16 && S > 1 && D > 0 && ldt >= 512 && ldt % 4 == 0 && ((uintptr_t)dtok & 15) == 0 && scale > 0.f,
                "mvd_gridattn_tokens_backward: bad shape (ldt >= 512, 16-byte aligned dtok)");
  MVD_CHECK_ARG(q0 >= 0 && Vq > 0 && q0 + Vq <= V, "mvd_gridattn_tokens_backward: bad query-view range");
  const size_t npts = (size_t)Vq * S * S * D;
  hipLaunchKernelGGL(tokens_bwd_kernel, dim3(cdiv(npts, 4)), dim3(256), 0, (hipStream_t)stream, x, depth_noise, steps, iter, grid_lin, cams,
                     in_cam, dtok, ldt, dfeat_acc, din_feat_acc, scale, V, q0, Vq, S, D, depth_scale, depth_shift);
  MVD_CHECK_LAUNCH("mvd_gridattn_tokens_backward");
  return 0;
}
