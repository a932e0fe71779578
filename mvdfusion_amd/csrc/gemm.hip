// mvd_gemm (include/mvd_hip.h): argument checks, kernel selection and launch, the split-K reduce kernels -- among them splitk_gn_kernel:
// reduce + epilogue + GroupNorm / SiLU of the output (optionally over its concatenation with a skip tensor) in one launch, a workgroup per
// (image, group) with the group's values in LDS -- and the weight / activation packing kernels.  The GEMM kernels themselves live in
// gemm_plain.hpp (gemm_kernel, one translation unit per block tile), gemm_ws.hip (role-split gemm_ws_kernel), gemm_patch.hip
// (conv_patch_kernel); the device code they share is gemm_device.hpp.
#include "gemm_device.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ split-K reduce
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const mvd_gemm_desc& d = p.d;
  const size_t MN = (size_t)d.M * d.N;
  if (d.epi == MVD_EPI_GEGLU) {
    const size_t total = (size_t)d.M * (d.N >> 1);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
      const int m = (int)(e / (d.N >> 1));
      const int col = (int)(e - (size_t)m * (d.N >> 1));
      const int pv = (col >> 4) * 32 + (col & 15);
      float v = 0.f, g = 0.f;
      for (int z = 0; z < p.splits; ++z) {
        v += d.workspace[z * MN + (size_t)m * d.N + pv];
        g += d.workspace[z * MN + (size_t)m * d.N + pv + 16];
      }
      epi_geglu_elem(d, m, pv, v, g);
    }
    return;
  }
  if (d.epi == MVD_EPI_STORE) {   // 4 columns per thread: 16-byte slab reads, vector epilogue
    const size_t MN4 = MN >> 2;
    const float inv = 1.0f / gemm_acc_scale(d);   // epi_store4 re-applies acc_scale; slabs hold raw accumulators
    (void)inv;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < MN4; e += (size_t)gridDim.x * 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int z = 0; z < p.splits; ++z) {
        const float4 t = *(const float4*)(d.workspace + z * MN + e * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      const int m = (int)((e * 4) / d.N);
      const int n = (int)(e * 4 - (size_t)m * d.N);
      if (n + 3 < d.n_store) {
        epi_store4(d, m, n, v);
      } else {
        epi_store_elem(d, m, n, v.x);
        epi_store_elem(d, m, n + 1, v.y);
        epi_store_elem(d, m, n + 2, v.z);
        epi_store_elem(d, m, n + 3, v.w);
      }
    }
    return;
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < MN; e += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < p.splits; ++z) v += d.workspace[z * MN + e];
    const int m = (int)(e / d.N);
    const int n = (int)(e - (size_t)m * d.N);
    epi_store_elem(d, m, n, v);
  }
}

// Split-K reduce for a GEMM whose output feeds a LayerNorm folded into its consumer (mvd_gemm_desc.rs_out): one wavefront per row and
// 256-column span; same sums and epilogue as splitk_reduce_kernel, then the row's {sum, sum of squares} over the span go to slot
// blockIdx.y of the row (wave reduction in a fixed order).  MVD_EPI_STORE, n_store == N.
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(GemmParams p) {
  const mvd_gemm_desc& d = p.d;
  const size_t MN = (size_t)d.M * d.N;
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.y * 256 + lane * 4;
  if (m >= d.M) return;
  float s1 = 0.f, q1 = 0.f;
  if (n < d.N) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* w = d.workspace + (size_t)m * d.N + n;
    for (int z = 0; z < p.splits; ++z) {
      const float4 t = *(const float4*)(w + z * MN);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float4 f = epi_store4(d, m, n, v);
    s1 = (f.x + f.y) + (f.z + f.w);
    q1 = (f.x * f.x + f.y * f.y) + (f.z * f.z + f.w * f.w);
  }
  s1 = wave_sum(s1);
  q1 = wave_sum(q1);
  if (lane == 0) *((float2*)d.rs_out + (size_t)m * d.rs_ld + blockIdx.y) = make_float2(s1, q1);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) d.rs_count[0] = gridDim.y;
}

// Split-K reduce for a GEMM whose output feeds a GroupNorm: one workgroup per (16-row slab, 256-column span); thread = (4 rows,
// one float4 column), same sums and epilogue as splitk_reduce_kernel.  The per-column {sum, sum of squares} of the slab are
// combined over the rows, then per group in column order (a group cut by the span boundary contributes from both workgroups),
// and added to the statistics of the consumer GroupNorm (gn_stats_add: integer atomics, order independent).  MVD_EPI_STORE,
// n_store == N.
template <int RPT>   // rows per thread: slabs of 4 * RPT rows (16, or 8 when the problem has too few slabs to fill the chip)
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(GemmParams p) {
  __shared__ float cs[2][4][256];
  const mvd_gemm_desc& d = p.d;
  const size_t MN = (size_t)d.M * d.N;
  const int m0 = blockIdx.x * (4 * RPT), n0 = blockIdx.y * 256;
  const int c4 = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = n0 + c4 * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < d.N) {
    float4 v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* w = d.workspace + (size_t)(m0 + rg * RPT) * d.N + n;
#pragma unroll 2
    for (int z = 0; z < p.splits; ++z) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const float4 t = *(const float4*)(w + z * MN + (size_t)r * d.N);
        v[r].x += t.x; v[r].y += t.y; v[r].z += t.z; v[r].w += t.w;
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = epi_value4(d, m0 + rg * RPT + r, n, v[r]);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const float4 f = v[r];
      epi_put4(d, m0 + rg * RPT + r, n, f);
      s.x += f.x; s.y += f.y; s.z += f.z; s.w += f.w;
      q.x += f.x * f.x; q.y += f.y * f.y; q.z += f.z * f.z; q.w += f.w * f.w;
    }
  }
  *(float4*)&cs[0][rg][c4 * 4] = s;
  *(float4*)&cs[1][rg][c4 * 4] = q;
  __syncthreads();
  const int c = threadIdx.x, nn = n0 + c;
  const float cs_ = cs[0][0][c] + cs[0][1][c] + cs[0][2][c] + cs[0][3][c];
  const float cq_ = cs[1][0][c] + cs[1][1][c] + cs[1][2][c] + cs[1][3][c];
  __syncthreads();
  cs[0][0][c] = cs_;
  cs[1][0][c] = cq_;
  __syncthreads();
  if (nn >= d.N) return;
  const int cg = d.N / d.gn_groups;
  const int g = nn / cg, pos = nn - g * cg;
  if (c != 0 && pos != 0) return;
  int len = cg - pos;
  if (len > 256 - c) len = 256 - c;
  float ss = 0.f, qq = 0.f;
  for (int j = 0; j < len; ++j) {
    ss += cs[0][0][c + j];
    qq += cs[1][0][c + j];
  }
  gn_stats_add(d.gn_stats, m0 / d.gn_hw, g, d.gn_groups, ss, qq);
}

// Split-K reduce + GroupNorm APPLY in one launch (mvd_gemm_desc.gna_out_sp): a workgroup per (image, group).  It sums the split-K slabs
// of its gn_hw x (N / groups) values, applies the STORE epilogue (scale, bias, per-image bias, residual), keeps the values in LDS, forms
// the group's mean / rstd (fp32 partial sums per thread in a fixed order, combined in double: deterministic) and writes the normalised,
// activated values as split planes -- what splitk_reduce_stats_kernel + gn_apply_stats(_cols)_kernel did in two launches with the fp32
// tensor making a round trip through memory in between (57 such pairs per configs[1] step).  The fp32 output is written unless the
// caller marks it unused; the statistics slot of the output still receives the sums (another consumer may normalise the same tensor).
// Mapping: two channels per thread (a group is N / 32 = 10, 20, 40 ... channels wide: 8-byte accesses); the 8 XCDs take runs of
// groups / 8 neighbouring groups each, so a 128-byte line of a slab row (3.2 groups of 10 channels) is pulled into one or two L2s.
#define MVD_GNK_THREADS 1024
__global__ __launch_bounds__(MVD_GNK_THREADS) void splitk_gn_kernel(GemmParams p) {
  // Latency, not bandwidth, is what this kernel has to manage: a workgroup owns a few thousand values spread over gn_hw rows, so every
  // thread takes U elements at a time and has ALL their operands -- up to ZC slabs, the residual, both biases -- in flight before it
  // touches one (a first version that walked its elements one by one was a chain of ~10 memory round trips per workgroup and slower than
  // the two kernels it replaces).  1024 threads: the largest group of a step (1024 x 10 channels) is five elements per thread.
  constexpr int NT = MVD_GNK_THREADS, NWV = NT / 64, U = 3, ZC = 4;
  extern __shared__ float s_val[];                 // [gn_hw][cg] values of the group
  __shared__ double s_red[2][NWV];
  __shared__ float s_coef[2][128];
  const mvd_gemm_desc& d = p.d;
  const int G = d.gn_groups, HW = d.gn_hw;
  // (concat mode, mvd_gemm_desc.cat_b: the GroupNorm runs over [out | cat_b], CT = N + cat_cb channels; channels >= N come from cat_b)
  const int CT = d.N + (d.cat_b ? d.cat_cb : 0);
  const int cg = CT / G, cg2 = cg >> 1;
  int g, b;
  {
    const int bid = blockIdx.x;
    if ((G & 7) == 0) {
      const int gpx = G >> 3, x = bid & 7, r = bid >> 3;
      g = x * gpx + r % gpx;
      b = r / gpx;
    } else {
      g = bid % G;
      b = bid / G;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = g * cg, m0 = b * HW;
  const size_t MN = (size_t)d.M * d.N;
  const int total = HW * cg2;
  const bool put_out = d.out != nullptr && !(d.gna_flags & MVD_GNA_OUT_UNUSED);
  // element e = (row e / cg2, channel pair e % cg2); a thread walks e = tid, tid + NT, ...: (r, j) advance without a division
  const int dq = NT / cg2, dj = NT - dq * cg2;
  const int rs = tid / cg2, js = tid - rs * cg2;                 // the thread's first element ...
  const int r0 = tid < total ? rs : 0, j0 = tid < total ? js : 0; // ... which is also its safe address for the unconditional loads
  const float* const zero2 = (const float*)g_zero_page;
  const bool has_bias = d.bias != nullptr, has_bb = d.bias_b != nullptr, has_res = d.res != nullptr;
  const float ascale = gemm_acc_scale(d);
  float s = 0.f, q = 0.f;
  {
    int r = rs, j = js;
    for (int e0 = tid; e0 < total; e0 += NT * U) {
      int rr[U], jj[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = e0 + u * NT < total;
        rr[u] = ok[u] ? r : r0;                    // (elements past the end re-read the thread's first one: every load is unconditional)
        jj[u] = ok[u] ? j : j0;
        r += dq;
        j += dj;
        if (j >= cg2) {
          j -= cg2;
          ++r;
        }
      }
      float2 acc[U], tb[U], tbb[U], tr[U];
      float2 t[ZC][U];
      const int nz0 = p.splits < ZC ? p.splits : ZC;
      bool own[U];                                  // the element is a column of THIS GEMM (else: of cat_b)
      float2 tc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = m0 + rr[u], nn = c0 + 2 * jj[u];
        own[u] = nn < d.N;
        const int n = own[u] ? nn : 0;              // (a cat_b element reads column 0 of the slabs and drops it: loads stay unconditional)
        tb[u] = *(const float2*)(has_bias ? d.bias + n : zero2);
        tbb[u] = *(const float2*)(has_bb ? d.bias_b + (size_t)(m / d.rows_per_batch) * d.ldbb + n : zero2);
        tr[u] = *(const float2*)(has_res ? d.res + (size_t)m * d.ldr + n : zero2);
        tc[u] = *(const float2*)(own[u] ? zero2 : d.cat_b + (size_t)m * d.cat_cb + (nn - d.N));
#pragma unroll
        for (int z = 0; z < ZC; ++z) {
          const float* w = d.workspace + (size_t)(z < nz0 ? z : 0) * MN + (size_t)m * d.N + n;
          t[z][u] = *(const float2*)w;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc[u] = make_float2(0.f, 0.f);
#pragma unroll
        for (int z = 0; z < ZC; ++z)
          if (z < nz0) {
            acc[u].x += t[z][u].x;
            acc[u].y += t[z][u].y;
          }
      }
      for (int zb = ZC; zb < p.splits; zb += ZC) {               // more than ZC slabs: further rounds of ZC x U loads
        const int nz = p.splits - zb < ZC ? p.splits - zb : ZC;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int m = m0 + rr[u], n = own[u] ? c0 + 2 * jj[u] : 0;
#pragma unroll
          for (int z = 0; z < ZC; ++z) t[z][u] = *(const float2*)(d.workspace + (size_t)(zb + (z < nz ? z : 0)) * MN + (size_t)m * d.N + n);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int z = 0; z < ZC; ++z)
            if (z < nz) {
              acc[u].x += t[z][u].x;
              acc[u].y += t[z][u].y;
            }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float2 v = acc[u];
        v.x = v.x * ascale + tb[u].x + tbb[u].x + tr[u].x;
        v.y = v.y * ascale + tb[u].y + tbb[u].y + tr[u].y;
        if (!own[u]) v = tc[u];
        if (ok[u]) {
          const int m = m0 + rr[u], n = c0 + 2 * jj[u];
          if (put_out && own[u]) *(float2*)(d.out + (size_t)m * d.ldo + n) = v;
          if (d.cat_raw_sp) {                      // raw planes of [out | cat_b] (the next ResBlock's 1x1 skip convolution reads them)
            uint32_t hh, ll;
            split_op16x2(v.x, v.y, hh, ll);
            u16* pp = (u16*)d.cat_raw_sp + sp_index((size_t)m, CT, n);
            *(uint32_t*)pp = hh;
            *(uint32_t*)(pp + 32) = ll;
          }
          *(float2*)(s_val + 2 * (e0 + u * NT)) = v;
          s += v.x + v.y;
          q += v.x * v.x + v.y * v.y;
        }
      }
    }
  }
  // Weight prefetch for the launches that follow (mvd_gemm_desc.pf_items, as in gemm_ws_kernel): this kernel's own loads have all been
  // consumed above and the rest of it (statistics, normalise, store) issues none, so the requests ride along for their issue slots; the
  // wave ends behind them (s_endpgm waits for outstanding memory instructions), by which time the stores of phase 2 are on their way too.
  unsigned pf_sink = 0;
  if (d.pf_items != nullptr) {
    const __attribute__((address_space(4))) mvd_prefetch_item* tab = (const __attribute__((address_space(4))) mvd_prefetch_item*)d.pf_items;
    const int gw = blockIdx.x * NWV + wave, GW = gridDim.x * NWV;
    for (int it = 0; it < d.pf_n; ++it) {
      const unsigned char* base = (const unsigned char*)tab[it].ptr;
      const long lines = (long)((tab[it].bytes + 127) >> 7);
      for (long l = (long)gw * 64 + lane; l < lines; l += (long)GW * 64) {
        const unsigned char* a = base + (l << 7);
        asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(a) : "memory");
      }
    }
  }
  {
    const double sd = wave_sum_d((double)s), qd = wave_sum_d((double)q);
    if (lane == 0) {
      s_red[0][wave] = sd;
      s_red[1][wave] = qd;
    }
  }
  __syncthreads();
  double S1 = 0.0, S2 = 0.0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) {
    S1 += s_red[0][w];
    S2 += s_red[1][w];
  }
  const double cnt = (double)HW * cg;
  const double mean = S1 / cnt;
  double var = S2 / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)d.gna_eps));
  if (tid < cg) {
    const float a = rstd * d.gna_gamma[c0 + tid];
    s_coef[0][tid] = a;
    s_coef[1][tid] = d.gna_beta[c0 + tid] - (float)mean * a;
  }
  if (tid == 0 && d.gn_stats) gn_stats_add(d.gn_stats, b, g, G, (float)S1, (float)S2);
  __syncthreads();
  u16* const ysp = (u16*)d.gna_out_sp;
  const int fl = d.gna_flags;
  {
    int r = rs, j = js;
    for (int e = tid; e < total; e += NT) {
      float2 v = *(const float2*)(s_val + 2 * e);
      v.x = v.x * s_coef[0][2 * j] + s_coef[1][2 * j];
      v.y = v.y * s_coef[0][2 * j + 1] + s_coef[1][2 * j + 1];
      if (fl & MVD_GNA_ROUND_F16) {
        v.x = (float)(_Float16)v.x;
        v.y = (float)(_Float16)v.y;
      }
      if (fl & MVD_GNA_SILU) {
        v.x = silu_f(v.x);
        v.y = silu_f(v.y);
      }
      uint32_t hh, ll;
      split_op16x2(v.x, v.y, hh, ll);
      u16* pp = ysp + sp_index((size_t)(m0 + r), CT, c0 + 2 * j);
      *(uint32_t*)pp = hh;
      *(uint32_t*)(pp + 32) = ll;
      r += dq;
      j += dj;
      if (j >= cg2) {
        j -= cg2;
        ++r;
      }
    }
  }
  asm volatile("" ::"v"(pf_sink));      // (the prefetch requests' landing register stays reserved to the end)
}

// Tile configurations (mvd_gemm_desc.cfg = 1 + MVD_GEMM_CFG_STRIDE * tile + 2 * loop + order; 0 = built-in heuristic).
//   tile : 0 = 64x64 (2x2 waves)  1 = 128x128 (2x4)  2 = 128x80 (4x1)  3 = 64x80 (4x1)  4 = 128x160 (4x2)
//          (a 256x128 tile -- 128x32 wave tiles, 64 MFMAs per k-tile and wave against 20 fragment reads and 6 DMAs -- was built and
//          measured in round 3: equal or slower on every shape of the step, profiles/r03_gemm_tile256_probe.log; dropped)
//   loop : 0 = plain two-buffer loop, 1 = register-pipelined loop, 2 = staggered wave groups, 3 LDS buffers (8-wave tiles 1 and
//          4 only), 4 = register-pipelined loop over a ring of <= 4 LDS buffers, 5 = over a ring of <= 8 (4-wave tiles 0, 2, 3 only: the
//          8-wave tiles fit 4), 6 = conv_patch_kernel (stride-1 3x3 convolutions, tiles 1, 2, 4), 7 = gemm_ws_kernel (consumer / loader
//          wavefronts, LDS-DMA delivery).
//          10 = REMOVED in round 6 (the persistent role-split kernel of round 5: bit-exact but slower than two co-resident plain workgroups and
//          never a tuner candidate; source and measurements live on under tools/probes/gemm_pt.hip, profiles/r05_pt_*).
//          3, 8, 9 = REMOVED in round 5 (staggered loop with four LDS buffers; register-staged delivery -- global_load -> VGPR ->
//          ds_write_b128 -- in gemm_ws_kernel and in gemm_kernel): built and measured in rounds 3 / 4 (profiles/r04_*), never selected by
//          the tuner on any shape of the step; mvd_gemm_cfg_supported() answers 0 for them and the numbering of the others is unchanged.
//   order : 0 = n-fastest tile order, 1 = m-fastest
// The 80-column family serves MVD_EPI_STORE only (the GEGLU / QKV epilogues walk a wave tile in 32-column blocks).
struct TileInfo {
  int bm, bn, waves;
  int cores_plain, cores_pipe;   // workgroups that fit one CU (LDS / registers), per loop variant
};
static const TileInfo kTiles[MVD_GEMM_TILES] = {{64, 64, 4, 5, 3}, {128, 128, 8, 2, 1}, {128, 80, 4, 2, 2}, {64, 80, 4, 4, 3},
                                                {128, 160, 8, 1, 1}};

// Split-K selection by a small time model (unit: 0.7 us ~ one DMA round trip).  What matters most is how evenly
// tiles*splits workgroups divide over the 256 CUs (192 tiles: 1, 2 or 3 splits all leave a CU with 180 k-tiles, 4 splits
// give every CU 3 x 45), then whether enough workgroups are co-resident to hide the per-k-tile DMA latency, then the cost
// of the fp32 partial-sum round trip.
static int choose_splits(long tiles, int nk, const TileInfo& ti, int loop, size_t mn) {
  const double area = (double)ti.bm * ti.bn / (128.0 * 128.0);
  const double t_mfma = 0.75 * area;                 // MFMA-pipe time of one k-tile of one workgroup
  const double t_lat = loop == 0 ? 1.0 : (loop == 1 ? 0.4 : (loop == 2 ? 0.2 : (loop == 5 ? 0.05 : 0.1)));      // (loops 3, 4, 6: 0.1)   // exposed DMA latency per k-tile, workgroup alone
  const double t_epi = 0.27 + 1.73 * area;
  const int coresident = loop == 0 ? ti.cores_plain : (loop == 1 ? ti.cores_pipe : (loop == 4 && ti.waves == 4 ? 2 : 1));      // (loop 7: 1)
  const double red_fixed = 6.0, red_per_split = (double)mn * 8.0 / 3.0e12 / 0.7e-6;
  int best = 1;
  double best_t = 1e30;
  const int smax = nk / 8 < 1 ? 1 : (nk / 8 > 32 ? 32 : nk / 8);
  for (int sp = 1; sp <= smax; ++sp) {
    const int iters = (nk + sp - 1) / sp;
    if ((long)iters * (sp - 1) >= nk) continue;      // an empty trailing split
    const long per_cu = (tiles * sp + 255) / 256;
    const long rounds = (per_cu + coresident - 1) / coresident;
    const double busy = (double)per_cu * (iters * t_mfma + t_epi);
    const double lat = (double)rounds * (iters * t_lat + t_epi);
    double t = busy > lat ? busy : lat;
    if (sp > 1) t += red_fixed + sp * red_per_split;
    if (t < best_t * 0.97) {                          // prefer fewer splits unless clearly better
      best_t = t;
      best = sp;
    }
  }
  return best;
}

// conv_patch_kernel serves stride-1, padded 3x3 convolutions whose BM-row tiles are whole image rows of one image or whole images;
// returns the patch DMAs per wave and k-tile (1 or 2), 0 when the problem does not fit.
static int patch_shares(const mvd_gemm_desc& d, const TileInfo& ti) {
  if (d.a_mode != MVD_A_CONV3X3 || d.b_mode != MVD_B_PACKED || d.stride != 1 || d.upsample || d.no_pad_tl) return 0;
  if (d.Hin != d.Hout || d.Win != d.Wout || ti.bm != 128) return 0;
  const int W = d.Wout, HW = d.Hout * d.Wout;
  if (ti.bm % W != 0 || (HW >= ti.bm ? HW % ti.bm != 0 : ti.bm % HW != 0)) return 0;
  const int nseg = HW >= ti.bm ? 1 : ti.bm / HW, Rb = HW >= ti.bm ? ti.bm / W : d.Hout;
  const int P = nseg * (Rb + 2) * (W + 2);
  if (P > MVD_PATCH_SLOTS_MAX) return 0;
  const int PG = (P + 7) / 8, cap = (11 - conv_patch_ring(ti.bn, ti.waves)) * ti.waves;
  return PG <= cap ? 1 : (PG <= 2 * cap ? 2 : 0);
}

static bool cfg_supported(const mvd_gemm_desc& d, int cfg) {
  if (cfg == 0) return true;
  if (cfg < 0 || cfg > MVD_GEMM_CFG_STRIDE * MVD_GEMM_TILES) return false;
  const int tile = (cfg - 1) / MVD_GEMM_CFG_STRIDE, loop = ((cfg - 1) % MVD_GEMM_CFG_STRIDE) >> 1;
  if (loop >= MVD_GEMM_LOOPS) return false;
  if (tile >= 2 && d.epi != MVD_EPI_STORE) return false;
  const int waves = kTiles[tile].waves;
  if ((loop == 2 || loop == 3) && waves != 8) return false;
  if (loop == 3 || loop >= 8) return false;                    // removed (never selected by the tuner): the four-buffer staggered loop, the two
                                                                // register-staged deliveries (round 5), the persistent role-split kernel (round 6);
                                                                // the numbering of the others is unchanged
  if (loop == 5 && waves != 4) return false;
  if (loop == 6) return (tile == 1 || tile == 2 || tile == 4) && patch_shares(d, kTiles[tile]) > 0;
  if (loop == 7) return tile == 1 || ((tile == 2 || tile == 4) && d.epi == MVD_EPI_STORE);   // (64x64 wave tiles: every epilogue)
  return true;
}

}  // namespace

extern "C" int mvd_gemm_cfg_supported(const mvd_gemm_desc* dp, int cfg) { return dp && cfg_supported(*dp, cfg) ? 1 : 0; }

extern "C" int mvd_gemm(const mvd_gemm_desc* dp, mvd_stream_t stream) {
  MVD_CHECK_ARG(dp != nullptr, "mvd_gemm: null descriptor");
  GemmParams p;
  p.d = *dp;
  mvd_gemm_desc& d = p.d;
  MVD_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "mvd_gemm: bad sizes M=%d N=%d K=%d", d.M, d.N, d.K);
  MVD_CHECK_ARG(d.K % 32 == 0, "mvd_gemm: K=%d must be a multiple of 32 (pad the packed weight)", d.K);
  MVD_CHECK_ARG(d.N % 16 == 0, "mvd_gemm: N=%d must be a multiple of 16 (pad the packed weight)", d.N);
  MVD_CHECK_ARG(d.prec == MVD_PREC_X1 || d.prec == MVD_PREC_X3 || d.prec == MVD_PREC_X4, "mvd_gemm: bad prec %d", d.prec);
  MVD_CHECK_ARG(d.A && d.Wp, "mvd_gemm: null operand");
  MVD_CHECK_ARG(((uintptr_t)d.A & 127) == 0 && ((uintptr_t)d.Wp & 127) == 0, "mvd_gemm: operands must be 128-byte aligned");
  if (d.a_mode == MVD_A_CONV3X3) {
    MVD_CHECK_ARG(d.Cin % 32 == 0 && d.K == 9 * d.Cin, "mvd_gemm: conv needs Cin %% 32 == 0 and K == 9*Cin (Cin=%d K=%d)", d.Cin, d.K);
    MVD_CHECK_ARG(d.M == d.B * d.Hout * d.Wout, "mvd_gemm: conv M mismatch");
    MVD_CHECK_ARG(d.stride == 1 || d.stride == 2, "mvd_gemm: conv stride must be 1 or 2");
    if (d.upsample) MVD_CHECK_ARG(d.stride == 1 && d.Hout == 2 * d.Hin && d.Wout == 2 * d.Win, "mvd_gemm: upsample geometry");
  } else {
    MVD_CHECK_ARG(d.a_mode == MVD_A_DENSE, "mvd_gemm: bad a_mode");
    MVD_CHECK_ARG(d.lda >= d.K && d.lda % 32 == 0, "mvd_gemm: lda=%d must be >= K=%d and a multiple of 32", d.lda, d.K);
  }
  if (d.out_sp) MVD_CHECK_ARG(d.ldp > 0 && d.ldp % 32 == 0 && ((uintptr_t)d.out_sp & 127) == 0,
                              "mvd_gemm: split-plane output needs ldp %% 32 == 0 and 128-byte alignment");
  if (d.out) MVD_CHECK_ARG(d.ldo % 4 == 0 && ((uintptr_t)d.out & 15) == 0, "mvd_gemm: out must be 16-byte aligned with ldo %% 4 == 0");
  if (d.res) MVD_CHECK_ARG(d.ldr % 4 == 0 && ((uintptr_t)d.res & 15) == 0, "mvd_gemm: res must be 16-byte aligned with ldr %% 4 == 0");
  MVD_CHECK_ARG(((uintptr_t)d.bias & 15) == 0 && ((uintptr_t)d.bias_b & 15) == 0 && ((uintptr_t)d.colscale & 15) == 0,
                "mvd_gemm: bias / bias_b / colscale must be 16-byte aligned");
  if (d.epi == MVD_EPI_STORE) {
    MVD_CHECK_ARG(d.out != nullptr || d.out_sp != nullptr, "mvd_gemm: no output");
    if (d.n_store <= 0 || d.n_store > d.N) d.n_store = d.N;
    if (d.bias_b) {
      MVD_CHECK_ARG(d.rows_per_batch > 0, "mvd_gemm: bias_b needs rows_per_batch");
      if (d.ldbb == 0) d.ldbb = d.N;
      MVD_CHECK_ARG(d.ldbb % 4 == 0, "mvd_gemm: ldbb %% 4 != 0");
    }
  } else if (d.epi == MVD_EPI_GEGLU) {
    MVD_CHECK_ARG((d.out != nullptr || d.out_sp != nullptr) && d.N % 32 == 0, "mvd_gemm: GEGLU needs an output and N %% 32 == 0");
    d.bias_b = nullptr;
  } else if (d.epi == MVD_EPI_QKV) {
    MVD_CHECK_ARG(d.q_hi && d.q_lo && d.k_hi && d.k_lo && d.vt_hi && d.vt_lo, "mvd_gemm: QKV planes missing");
    MVD_CHECK_ARG(d.heads > 0 && d.dhead > 0 && d.N == 3 * d.heads * d.dhead, "mvd_gemm: QKV needs N == 3*heads*dhead");
    MVD_CHECK_ARG(d.L > 0 && d.M % d.L == 0 && d.Lpad >= d.L, "mvd_gemm: QKV needs M %% L == 0");
    MVD_CHECK_ARG(d.dhead % 4 == 0 && d.L % 4 == 0, "mvd_gemm: QKV needs dhead %% 4 == 0 and L %% 4 == 0");
  } else {
    MVD_CHECK_ARG(false, "mvd_gemm: bad epilogue %d", d.epi);
  }
  if (d.rs_out)
    MVD_CHECK_ARG(d.epi == MVD_EPI_STORE && d.n_store == d.N && !d.gn_stats && d.rs_count && d.rs_ld >= cdiv(d.N, 32) && ((uintptr_t)d.rs_out & 7) == 0,
                  "mvd_gemm: rs_out needs MVD_EPI_STORE, n_store == N, no gn_stats, rs_count and rs_ld >= N / 32 (N=%d rs_ld=%d)", d.N, d.rs_ld);
  if (d.ln_stats) {
    MVD_CHECK_ARG((d.epi == MVD_EPI_QKV || d.epi == MVD_EPI_GEGLU) && d.ln_count && d.ln_colsum && d.ln_dim > 0 && d.ln_ld > 0 &&
                      ((uintptr_t)d.ln_stats & 7) == 0 && ((uintptr_t)d.ln_colsum & 15) == 0,
                  "mvd_gemm: ln_stats (LayerNorm fold) serves the QKV / GEGLU epilogues and needs ln_count, ln_colsum, ln_dim, ln_ld");
    d.splitk = 1;      // the fold lives in the tile epilogue (the row statistics are per tile row)
  }
  if (d.gn_stats)
    MVD_CHECK_ARG(d.epi == MVD_EPI_STORE && d.n_store == d.N && d.M % 16 == 0 && d.gn_hw > 0 && d.gn_hw % 16 == 0 && d.gn_groups > 0 &&
                      d.N % d.gn_groups == 0,
                  "mvd_gemm: gn_stats needs MVD_EPI_STORE, n_store == N, M %% 16 == 0, gn_hw %% 16 == 0, N %% gn_groups == 0 (N=%d M=%d hw=%d)",
                  d.N, d.M, d.gn_hw);
  if (d.gna_out_sp)
    MVD_CHECK_ARG(d.gn_stats && d.out && d.ldo == d.N && d.gna_gamma && d.gna_beta && d.N % 32 == 0 && d.M % d.gn_hw == 0 &&
                      ((uintptr_t)d.gna_out_sp & 127) == 0,
                  "mvd_gemm: gna_out_sp (GroupNorm apply behind the GEMM) needs gn_stats, out with ldo == N, gamma / beta, N %% 32 == 0, M %% gn_hw == 0");
  if (d.cat_b)
    MVD_CHECK_ARG(d.gna_out_sp && d.cat_cb > 0 && d.cat_cb % 2 == 0 && (d.N + d.cat_cb) % 32 == 0 && (d.N + d.cat_cb) % d.gn_groups == 0 &&
                      ((uintptr_t)d.cat_b & 7) == 0 && ((uintptr_t)d.cat_raw_sp & 127) == 0 &&
                      mvd_concat_groupnorm_fits(d.N, d.cat_cb, d.gn_hw, d.gn_groups),
                  "mvd_gemm: cat_b (GroupNorm over [out | cat_b]) needs gna_out_sp, an even cat_cb, (N + cat_cb) %% 32 == 0 and a shape "
                  "mvd_concat_groupnorm_fits() accepts (N=%d cat_cb=%d hw=%d)", d.N, d.cat_cb, d.gn_hw);
  long long* const gna_stats = d.gn_stats;       // the statistics slot of the tensor the GroupNorm normalises ...
  if (d.cat_b) d.gn_stats = nullptr;             // ... which in concat mode is [out | cat_b]: the GEMM's own epilogue / reduce must not touch it
  if (d.b_mode == MVD_B_PLANES)
    MVD_CHECK_ARG(d.ldb >= d.K && d.ldb % 32 == 0, "mvd_gemm: B planes need ldb=%d >= K=%d, a multiple of 32", d.ldb, d.K);
  else
    MVD_CHECK_ARG(d.b_mode == MVD_B_PACKED, "mvd_gemm: bad b_mode %d", d.b_mode);
  if (d.acc_scale == 0.f) d.acc_scale = 1.f;
  p.nk = d.K / 32;
  p.nt16 = d.N / 16;
  // ---- kernel configuration: explicit (cfg >= 1) or the built-in heuristic (128x128 once the grid fills the chip, else 64x64)
  int tile, loop = 0, order = -1;
  MVD_CHECK_ARG(d.cfg >= 0 && d.cfg <= MVD_GEMM_CFG_STRIDE * MVD_GEMM_TILES, "mvd_gemm: bad cfg %d", d.cfg);
  if (d.cfg >= 1) {
    tile = (d.cfg - 1) / MVD_GEMM_CFG_STRIDE;
    loop = ((d.cfg - 1) % MVD_GEMM_CFG_STRIDE) >> 1;
    order = (d.cfg - 1) & 1;
    MVD_CHECK_ARG(cfg_supported(d, d.cfg), "mvd_gemm: cfg %d (tile %d, loop %d) does not serve this problem (include/mvd_hip.h: cfg)", d.cfg, tile,
                  loop);
  } else {
    const long tiles128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
    tile = (tiles128 >= 128 && (d.N >= 512 || d.K >= 2048)) ? 1 : 0;
  }
  const TileInfo& ti = kTiles[tile];
  p.tiles_n = cdiv(d.N, ti.bn);
  p.tiles_m = cdiv(d.M, ti.bm);
  if (order < 0) {
    // bytes each XCD pulls through its L2 under the two tile orders (8 XCDs, operands are 4 B per element)
    const double W = (double)d.N * d.K * 4.0;
    const double A = (double)d.M * d.K * 4.0 / (d.a_mode == MVD_A_CONV3X3 ? 9.0 : 1.0);
    const double rep_m = p.tiles_m < 8 ? p.tiles_m : 8, rep_n = p.tiles_n < 8 ? p.tiles_n : 8;
    order = (W + A * rep_n) < (W * rep_m + A) ? 1 : 0;
  }
  p.m_fastest = order;
  int splits = d.splitk;
  const long tiles = (long)p.tiles_m * p.tiles_n;
  if (splits == 0) splits = choose_splits(tiles, p.nk, ti, loop, (size_t)d.M * d.N);
  if (splits < 1) splits = 1;
  if (splits > p.nk) splits = p.nk;
  if (splits > 1) {
    if (d.workspace == nullptr) splits = 1;
    else {
      const size_t cap = d.workspace_elems / ((size_t)d.M * d.N);
      if ((size_t)splits > cap) splits = cap < 1 ? 1 : (int)cap;
    }
  }
  p.kt_per_split = cdiv(p.nk, splits);
  if (loop == 6) p.kt_per_split = 9 * cdiv(p.nk / 9, splits);       // conv_patch_kernel: a split is a run of whole channel blocks
  p.splits = cdiv(p.nk, p.kt_per_split);
  hipStream_t s = (hipStream_t)stream;
  bool launched;
  if (loop == 6) launched = mvd_gemm_launch_patch(tile, p, s, patch_shares(d, ti));
  else if (loop == 7) launched = mvd_gemm_launch_ws(tile, p, s);
  else if (tile == 0) launched = mvd_gemm_launch_plain_t0(loop, p, s);
  else if (tile == 1) launched = mvd_gemm_launch_plain_t1(loop, p, s);
  else if (tile == 2) launched = mvd_gemm_launch_plain_t2(loop, p, s);
  else if (tile == 3) launched = mvd_gemm_launch_plain_t3(loop, p, s);
  else launched = mvd_gemm_launch_plain_t4(loop, p, s);
  MVD_CHECK_ARG(launched, "mvd_gemm: no kernel for tile %d loop %d", tile, loop);
  MVD_CHECK_LAUNCH("mvd_gemm");
  // GroupNorm apply behind the GEMM (gna_out_sp): one reduce + apply kernel when the (image, group) slab of a split GEMM fits the LDS,
  // else the ordinary producer statistics followed by the apply kernel
  const int gna_ct = d.N + (d.cat_b ? d.cat_cb : 0);
  const int gna_cg = d.gna_out_sp ? gna_ct / d.gn_groups : 0;
  const size_t gna_lds = (size_t)d.gn_hw * gna_cg * 4;
  const bool gna_fused = d.gna_out_sp && p.splits > 1 && !d.rs_out && !d.out_sp && !d.colscale && d.act == MVD_ACT_NONE && (gna_cg & 1) == 0 &&
                         gna_cg <= 128 && gna_lds <= 128 * 1024 && d.ldo % 2 == 0 && (!d.res || d.ldr % 2 == 0) && (!d.bias_b || d.ldbb % 2 == 0) &&
                         d.N % 2 == 0;
  if (gna_fused) {
    {                  // (more than the default 64 KiB of dynamic LDS: 1024 rows x 30 channels of a concatenation)
      static unsigned long long raised = 0;
      const hipError_t e = mvd_raise_dynamic_lds((const void*)splitk_gn_kernel, 128 * 1024, &raised);
      MVD_CHECK_ARG(e == hipSuccess, "mvd_gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
    }
    p.d.gn_stats = gna_stats;      // (the slot of the normalised tensor -- of the concatenation in concat mode)
    if (loop == 7) p.d.pf_items = nullptr;      // (the role-split kernel's consumer wavefronts already requested this launch's prefetch share)
    hipLaunchKernelGGL(splitk_gn_kernel, dim3((d.M / d.gn_hw) * d.gn_groups), dim3(MVD_GNK_THREADS), gna_lds, s, p);
    MVD_CHECK_LAUNCH("mvd_gemm/splitk_gn");
    return 0;
  }
  if (p.splits > 1) {
    if (d.rs_out) {
      hipLaunchKernelGGL(splitk_reduce_rows_kernel, dim3(cdiv(d.M, 4), cdiv(d.N, 256)), dim3(256), 0, s, p);
    } else if (d.gn_stats) {
      const int spans = (d.N + 255) / 256;
      if ((d.M / 16) * spans >= 1024)
        hipLaunchKernelGGL(splitk_reduce_stats_kernel<4>, dim3(d.M / 16, spans), dim3(256), 0, s, p);
      else
        hipLaunchKernelGGL(splitk_reduce_stats_kernel<2>, dim3(d.M / 8, spans), dim3(256), 0, s, p);
    } else {
      const size_t total = (size_t)d.M * d.N;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
    }
    MVD_CHECK_LAUNCH("mvd_gemm/splitk_reduce");
  }
  if (d.gna_out_sp && d.cat_b)     // concat mode without the fused reduce: the output just written + cat_b -> one concat-and-normalise launch
    return mvd_concat_groupnorm(d.out, d.N, d.cat_b, d.cat_cb, nullptr, d.cat_raw_sp, d.gna_out_sp, d.gna_gamma, d.gna_beta, gna_stats,
                                d.M / d.gn_hw, d.gn_hw, d.gn_groups, d.gna_eps, d.gna_flags & (MVD_GNA_SILU | MVD_GNA_ROUND_F16), stream);
  if (d.gna_out_sp)
    return mvd_groupnorm_from_stats(d.out, d.gna_out_sp, d.gna_gamma, d.gna_beta, d.gn_stats, d.M / d.gn_hw, d.gn_hw, d.N, d.gn_groups,
                                    d.gna_eps, d.gna_flags & (MVD_GNA_SILU | MVD_GNA_ROUND_F16), stream);
  return 0;
}

// ------------------------------------------------------------------------------------------------ packing
namespace {
// One thread per 16-byte chunk of a fragment image: lane l of a wavefront fills lane l of one 16 (n) x 32 (k) micro-tile -- row n = 16 nt +
// (l & 15), k = 32 kt + 8 (l >> 4) ... + 7 -- so a wavefront reads 16 rows x 128 bytes and writes the micro-tile's hi image and lo image as
// two contiguous 1 KiB stores (round 6; it was one thread per element with 2-byte scattered stores: the weight re-pack after an optimizer
// step, 886 launches, cost the training step 14.5 ms).
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, u16* __restrict__ out, int N, int K,
                                                   int Np, int Kp, int ldw, int geglu, int conv_cin, int trans,
                                                   float scale) {
  const int nt16 = Np >> 4;
  const size_t total = (size_t)(Kp >> 5) * nt16 * 64;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t mt = e >> 6;                          // micro-tile: kt * nt16 + nt
    const int l = (int)(e & 63);
    const int kt = (int)(mt / nt16), nt = (int)(mt - (size_t)kt * nt16);
    const int n = nt * 16 + (l & 15);                  // packed row
    const int k = kt * 32 + (l >> 4) * 8;
    int src_n = n;
    if (geglu) {
      const int blk = n >> 5, within = n & 31;
      src_n = within < 16 ? blk * 16 + within : (N >> 1) + blk * 16 + (within - 16);
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (src_n < N) {
      if (conv_cin > 0) {
        const int cb = kt / 9, tap = kt - cb * 9;      // K order: (32-channel block, tap, channel in block)
        const int ci0 = cb * 32 + (k & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (ci0 + j < conv_cin) v[j] = w[((size_t)src_n * conv_cin + ci0 + j) * 9 + tap];
      } else if (trans) {                              // the source is the TRANSPOSE: (K, N) row-major (dgrad weights: no torch copy)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k + j < K) v[j] = w[(size_t)(k + j) * ldw + src_n];
      } else {
        const float* src = w + (size_t)src_n * ldw + k;
        if (k + 7 < K && (ldw & 3) == 0 && ((uintptr_t)w & 15) == 0) {
          const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (k + j < K) v[j] = src[j];
        }
      }
    }
    // [kt][nt][hi image | lo image]; an image is the 16x16x32 MFMA B fragment of the micro-tile as the wave holds it: lane
    // l = (n & 15) + 16 * (k-chunk of 8) owns 16 contiguous bytes -- one fully coalesced 1 KiB access per image
    // (the LDS-DMA source of a granule is contiguous, and the fragment reads from LDS are lane-contiguous: no bank conflicts)
    union { uint4 q; u16 h[8]; } H, Lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) split_op16(v[j] * scale, H.h[j], Lo.h[j]);
    u16* o = out + mt * 1024 + l * 8;
    *(uint4*)o = H.q;
    *(uint4*)(o + 512) = Lo.q;
  }
}

// fp32 (rows, cols) with leading dim ldx -> split planes (rows, ldp); columns [cols, ldp) are zero filled
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, u16* __restrict__ sp, size_t rows, int cols,
                                                           int ldx, int ldp, const float* __restrict__ scale) {
  const float sc = scale != nullptr ? *scale : 1.f;
  const int c4 = ldp >> 2;
  const size_t total = rows * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t r = e / c4;
    const int c = (int)(e - r * c4) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (c + j) < cols ? x[r * ldx + c + j] * sc : 0.f;
    store_sp4(sp, r, ldp, c, v[0], v[1], v[2], v[3]);
  }
}
}  // namespace

extern "C" size_t mvd_packed_weight_bytes(int N, int K) {
  const size_t Np = (size_t)((N + 15) & ~15), Kp = (size_t)((K + 31) & ~31);
  return Np * Kp * 4;
}

extern "C" int mvd_operand_format(void) { return MVD_OPERAND_FORMAT; }

extern "C" int mvd_pack_linear_weight(const float* w, int N, int K, int ldw, int geglu, float scale, void* packed,
                                      mvd_stream_t stream) {
  MVD_CHECK_ARG(w && packed && N > 0 && K > 0 && ldw >= K, "mvd_pack_linear_weight: bad arguments");
  if (geglu) MVD_CHECK_ARG(N % 32 == 0, "mvd_pack_linear_weight: geglu needs N %% 32 == 0");
  const int Np = (N + 15) & ~15, Kp = (K + 31) & ~31;
  const size_t total = (size_t)Np * Kp / 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (u16*)packed, N, K, Np, Kp, ldw,
                     geglu, 0, 0, scale);
  MVD_CHECK_LAUNCH("mvd_pack_linear_weight");
  return 0;
}

extern "C" int mvd_pack_linear_weight_t(const float* wt, int N, int K, int ldw, float scale, void* packed, mvd_stream_t stream) {
  MVD_CHECK_ARG(wt && packed && N > 0 && K > 0 && ldw >= N, "mvd_pack_linear_weight_t: bad arguments");
  const int Np = (N + 15) & ~15, Kp = (K + 31) & ~31;
  const size_t total = (size_t)Np * Kp / 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, wt, (u16*)packed, N, K, Np, Kp, ldw, 0, 0, 1, scale);
  MVD_CHECK_LAUNCH("mvd_pack_linear_weight_t");
  return 0;
}

extern "C" int mvd_pack_conv3x3_weight(const float* w, int Cout, int Cin, int cin_pad, float scale, void* packed,
                                       mvd_stream_t stream) {
  MVD_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && cin_pad >= Cin && cin_pad % 32 == 0,
                "mvd_pack_conv3x3_weight: bad arguments (cin_pad must be a multiple of 32)");
  const int Np = (Cout + 15) & ~15, Kp = 9 * cin_pad;
  const size_t total = (size_t)Np * Kp / 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (u16*)packed, Cout, Kp, Np, Kp, 0,
                     0, Cin, 0, scale);
  MVD_CHECK_LAUNCH("mvd_pack_conv3x3_weight");
  return 0;
}

extern "C" int mvd_split_planes_scaled(const float* x, void* sp, size_t rows, int cols, int ldx, int ldp, const float* scale_dev,
                                       mvd_stream_t stream) {
  MVD_CHECK_ARG(x && sp && rows > 0 && cols > 0 && ldx >= cols && ldp >= cols && ldp % 32 == 0,
                "mvd_split_planes: bad arguments (ldp must be a multiple of 32)");
  const size_t total = rows * (size_t)(ldp / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (u16*)sp, rows, cols, ldx, ldp, scale_dev);
  MVD_CHECK_LAUNCH("mvd_split_planes");
  return 0;
}

extern "C" int mvd_split_planes(const float* x, void* sp, size_t rows, int cols, int ldx, int ldp, mvd_stream_t stream) {
  return mvd_split_planes_scaled(x, sp, rows, cols, ldx, ldp, nullptr, stream);
}
