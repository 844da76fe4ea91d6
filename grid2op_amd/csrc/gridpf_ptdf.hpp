// gridpf_ptdf.hpp -- DC sensitivity path: p_or = PTDF * P_bus for every lane of a batch with ONE topology.
//
// The reference solves B' theta = P from scratch on every DC power flow (pp.rundcpp, grid2op/Backend/
// pandaPowerBackend.py:1090).  For a fixed topology the DC branch flows are LINEAR in the bus injections, so the
// factorisation is done once (host, gpf_ptdf_build) and a batch is evaluated as one GEMM
//     flow[lane][line] = sum_bus  P_bus[lane][bus] * PTDFt[bus][line]
// -- the one GEMM-shaped operation of this code base, on the FP64 matrix cores (v_mfma_f64_16x16x4_f64; the lane
// layout of its operands / results was probed on gfx950 with tools/mfma_probe.hip).
#pragma once
#include "gridpf_common.hpp"

namespace gpf {

typedef double v4d __attribute__((ext_vector_type(4)));

struct PtdfDev {
  int n_inj, nb_pad, line_pad, n_line;
  const int* inj_bus;        // [n_inj] COMPACT index (active buses of the PTDF topology only) of the bus of the injection
                             // column, -1: not an active-power injection
  const double* inj_w;       // [n_inj] weight (+1 generators except the slack, -1 loads / storages, -factor shunts)
  const double* ptdf_t;      // [nb_pad][line_pad] row-major: transposed PTDF over the compact active buses (zero rows /
                             // columns for padding, reference buses, open lines)
  // ---- per-lane topologies (gpf_ptdf_build_batch, gridpf_ptdf_batch.hpp); order == nullptr: ONE topology for all lanes --------------
  const int* order;          // [n_slots] slot -> lane, lanes grouped by topology class, every group padded to a multiple of 16 with -1
  const int* blk_class;      // [n_slots / 16] class of each group of 16 slots
  const int* cls_desc;       // class descriptors; the class's inj_bus table starts at cls_desc + cls * desc_stride + inj_bus_off
  const int* cls_status;     // [n_classes] != 0: no sensitivities for this class (islanded / singular): its lanes get NaN flows
  int desc_stride, inj_bus_off;
  long long ptdf_stride;     // doubles between the PTDF^T blocks of two classes
};

// K_PG: bus injections + GEMM in ONE launch.  Block (x, y) of 4 wavefronts: 16 lanes x the line tiles 4y .. 4y+3 (one 16 x 16
// tile per wavefront).  Prologue: the block builds the A operand -- the bus active-power injections P_bus[16][nb] of its 16
// lanes (generators except the slack minus loads, storages and shunt conductances) -- in LDS with f64 atomics straight from the
// lanes' injection rows (coalesced: consecutive threads read consecutive injection columns); it never goes through HBM.  The row
// stride is odd so that the 16 rows an MFMA operand read touches fall on different banks.
// MFMA operand layout (64 lanes, l = lane id): A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16],
// D[i = 4 * v + l / 16][j = l % 16] for the 4 result registers v.
__host__ __device__ inline int ptdf_a_stride(int nb_pad) { return nb_pad | 1; }
__global__ __launch_bounds__(256) void ptdf_flows_kernel(PtdfDev P, const double* __restrict__ inj, int lane0, int n_lanes,
                                                         float* __restrict__ flow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* A = reinterpret_cast<double*>(smem);             // [16][stride]
  const int stride = ptdf_a_stride(P.nb_pad);
  const int row0 = blockIdx.x * 16;                        // first lane (of the range) / first slot of this block
  __shared__ int s_lane[16];
  const int* inj_bus = P.inj_bus;
  const double* ptdf_t = P.ptdf_t;
  bool nan_out = false;
  if (P.order) {                                           // per-lane topologies: the 16 slots of a block share one class
    const int cls = P.blk_class[blockIdx.x];
    inj_bus = P.cls_desc + (size_t)cls * P.desc_stride + P.inj_bus_off;
    ptdf_t += (size_t)cls * P.ptdf_stride;
    nan_out = P.cls_status[cls] != 0;
    if (threadIdx.x < 16) s_lane[threadIdx.x] = P.order[row0 + threadIdx.x];
  } else if (threadIdx.x < 16) s_lane[threadIdx.x] = (row0 + (int)threadIdx.x < n_lanes) ? lane0 + row0 + (int)threadIdx.x : -1;
  for (int i = threadIdx.x; i < 16 * stride; i += 256) A[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < P.n_inj; i += 256) {
    const int b = inj_bus[i];
    if (b < 0) continue;
    const double w = P.inj_w[i];
    double v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (s_lane[r] >= 0) ? inj[(size_t)s_lane[r] * P.n_inj + i] : 0.0;      // 16 loads in flight
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(&A[r * stride + b], v[r] * w);
  }
  __syncthreads();
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ksteps = P.nb_pad / 4;
  const double* arow = A + (l & 15) * stride + (l >> 4);
  const int n_tiles = P.line_pad / 16;
  for (int t = blockIdx.y * 4 + w; t < n_tiles; t += 4 * gridDim.y) {      // one 16 x 16 tile per wavefront when gridDim.y covers the tiles
    v4d c = {0.0, 0.0, 0.0, 0.0};
    const double* bcol = ptdf_t + (size_t)(l >> 4) * P.line_pad + t * 16 + (l & 15);
    int s = 0;
    for (; s + 8 <= ksteps; s += 8) {                       // 8 k-steps per trip: 8 L2 loads + 8 LDS reads in flight, then 8 MFMAs
      double a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = arow[4 * (s + u)]; b[u] = bcol[(size_t)4 * (s + u) * P.line_pad]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], c, 0, 0, 0);
    }
    for (; s < ksteps; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * s], bcol[(size_t)4 * s * P.line_pad], c, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int ln = s_lane[4 * v + (l >> 4)];
      if (ln >= 0) flow[(size_t)ln * P.line_pad + t * 16 + (l & 15)] = nan_out ? __builtin_nanf("") : (float)c[v];
    }
  }
}

// K_PR: the same GEMM over M = lanes x chronics rows.  A batch of 2 048 lanes x ONE row is 94 MFLOP -- a few microseconds, launch /
// latency bound (13.6 % of the FP64-MFMA peak in round 3); the flows of T consecutive chronics rows for the same topology are ONE GEMM
// with M = lanes x T.  Pair p = row * n_lanes + lane; a block owns PTDF_MT * 16 pairs and ALL line tiles (wavefront w: tiles w, w + 4,
// ...): its prologue gathers the injections of its pairs straight from the device-resident chronics table as K9 of the step kernel
// does -- lane k reads row (t0 + row + lane_offset[k]) mod T of table lane_table[k], loads x lane_scale, non-slack prod_p rescaled to
// rebalance x sum(load) (float32, Environment/baseEnv.py:2516-2563 feeds these vectors), + the lane's redispatch delta; storage and
// shunt set-points from the lane's injection row -- and builds P_bus[pairs][bus] in LDS (f64 atomics); every B operand fetched from
// L2 then feeds PTDF_MT MFMAs (4 x the arithmetic per PTDF^T byte of K_PG) and the chronics gather is done once per pair.
struct PtdfRowsDev {
  const float* chron;          // [n_tab][T][n_chron]: load_p | load_q | prod_p | prod_v
  const int* lane_table;       // [lanes]
  const int* lane_offset;      // [lanes]
  const float* lane_scale;     // [lanes][2 n_load] or nullptr
  const float* lane_gen_delta; // [lanes][n_gen] or nullptr
  const unsigned char* gen_slack;
  int T, n_chron, n_load, n_gen, inj_gen_p, inj_load_p, inj_sto_p, n_inj_tail;   // n_inj_tail: injection columns from inj_sto_p on (storage / shunt set-points)
  int kpad;                    // rows of ptdf_t rounded up to a multiple of 32 (zero rows behind nb_pad): whole trips of 8 k-steps
  double rebalance;            // <= 0: prod_p as in the table
};
__host__ __device__ inline int ptdf_rows_stride(int kpad) { return kpad | 1; }
constexpr int PTDF_EPT = 8;    // chronics values a gather thread keeps in registers per kind (loads, generators)
template <int MT>
__global__ __launch_bounds__(256) void ptdf_rows_kernel(PtdfDev P, PtdfRowsDev R, const double* __restrict__ inj, int n_lanes, long long lane_stride,
                                                        int t0, int n_rows, float* __restrict__ flow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NP = 16 * MT;                               // pairs per block
  constexpr int TPP = 256 / NP;                             // threads per pair in the gather (4 / 8 / 16)
  double* A = reinterpret_cast<double*>(smem);              // [NP][stride]
  const int stride = ptdf_rows_stride(R.kpad);
  const long long n_pairs = (long long)n_lanes * n_rows;
  const long long p0 = (long long)blockIdx.x * NP;
  for (int i = threadIdx.x; i < NP * stride; i += 256) A[i] = 0.0;
  // ---- gather: TPP threads per pair, 4 elements in flight per thread -------------------------------------------------------------
  // (chronics row, lane) of the block's pairs: ONE 64-bit division per block (uniform), 32-bit arithmetic per pair, kept in LDS for
  // the epilogue (a 64-bit division per stored element cost more than the whole GEMM)
  __shared__ int p_row[NP], p_lane[NP];
  const int* inj_bus = P.inj_bus;
  const double* ptdf_t = P.ptdf_t;
  bool nan_out = false;
  if (P.order) {
    // per-lane topologies: n_lanes = number of SLOTS (a multiple of 16); block -> (group of 16 slots sg, MT consecutive rows): the
    // MT row tiles of the block are the same 16 lanes at MT chronics rows, so that they share the class's PTDF^T operands
    const int n_sg = n_lanes / 16, sg = blockIdx.x % n_sg, rg = blockIdx.x / n_sg;
    const int cls = P.blk_class[sg];
    inj_bus = P.cls_desc + (size_t)cls * P.desc_stride + P.inj_bus_off;
    ptdf_t += (size_t)cls * P.ptdf_stride;
    nan_out = P.cls_status[cls] != 0;
    if (threadIdx.x < NP) {
      const int m_ = threadIdx.x / 16, rw_ = rg * MT + m_, ln_ = P.order[sg * 16 + (threadIdx.x & 15)];
      p_row[threadIdx.x] = (rw_ < n_rows && ln_ >= 0) ? rw_ : -1;
      p_lane[threadIdx.x] = ln_ >= 0 ? ln_ : 0;
    }
  } else {
    const long long row0 = p0 / n_lanes;
    const unsigned ln0 = (unsigned)(p0 - row0 * n_lanes);
    if (threadIdx.x < NP) {
      const unsigned x = ln0 + threadIdx.x, dr = x / (unsigned)n_lanes;
      const bool ok = p0 + threadIdx.x < n_pairs;
      p_row[threadIdx.x] = ok ? (int)(row0 + dr) : -1;
      p_lane[threadIdx.x] = (int)(x - dr * (unsigned)n_lanes);
    }
  }
  __syncthreads();
  const int r = threadIdx.x / TPP, q = threadIdx.x % TPP;
  const bool on = p_row[r] >= 0;
  const int row = on ? p_row[r] : 0, lane = on ? p_lane[r] : 0;
  int crow = (t0 + row + (R.lane_offset ? R.lane_offset[lane] : 0)) % R.T;
  if (crow < 0) crow += R.T;
  const float* ch = R.chron + ((size_t)(R.lane_table ? R.lane_table[lane] : 0) * R.T + crow) * R.n_chron;
  const float* sc = R.lane_scale ? R.lane_scale + (size_t)lane * 2 * R.n_load : nullptr;
  // ONE pass over the chronics row: every thread keeps its (at most PTDF_EPT) load and generator values in registers -- all the
  // global loads of the gather are in flight together --, the TPP threads of a pair reduce the two sums of the rebalancing rule,
  // then the values go to P_bus with LDS atomics.  (Rows with more than PTDF_EPT * TPP loads or generators: the tail is re-read.)
  constexpr int EPT = PTDF_EPT;
  float lv[EPT], gv[EPT];
  int lb[EPT], gb[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    const int i = q + u * TPP;
    const bool okl = on && i < R.n_load, okg = on && i < R.n_gen;
    lv[u] = okl ? ch[i] * (sc ? sc[i] : 1.f) : 0.f;
    lb[u] = okl ? inj_bus[R.inj_load_p + i] : -1;
    gv[u] = okg ? ch[2 * R.n_load + i] : 0.f;
    gb[u] = okg ? inj_bus[R.inj_gen_p + i] : -2;          // -1: slack generator or not connected, -2: no such generator
  }
  double s_load = 0.0, s_prod = 0.0;
  if (R.rebalance > 0.0) {
#pragma unroll
    for (int u = 0; u < EPT; ++u) { s_load += (double)lv[u]; if (q + u * TPP < R.n_gen && !R.gen_slack[q + u * TPP]) s_prod += (double)gv[u]; }
    for (int i = q + EPT * TPP; i < R.n_load; i += TPP) s_load += (double)(ch[i] * (sc ? sc[i] : 1.f));
    for (int i = q + EPT * TPP; i < R.n_gen; i += TPP) if (!R.gen_slack[i]) s_prod += (double)ch[2 * R.n_load + i];
#pragma unroll
    for (int o = 1; o < TPP; o <<= 1) { s_load += __shfl_xor(s_load, o); s_prod += __shfl_xor(s_prod, o); }
  }
  const float sp = (R.rebalance > 0.0 && s_prod > 0.0) ? (float)(R.rebalance * s_load / s_prod) : 1.0f;
  __syncthreads();                                          // A is zero
  if (on) {
    double* Ar = A + (size_t)r * stride;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int i = q + u * TPP;
      if (lb[u] >= 0) atomicAdd(&Ar[lb[u]], (double)lv[u] * P.inj_w[R.inj_load_p + i]);
      if (gb[u] >= 0) {
        float pp = gv[u] * sp;
        if (R.lane_gen_delta) pp += R.lane_gen_delta[(size_t)lane * R.n_gen + i];
        atomicAdd(&Ar[gb[u]], (double)pp * P.inj_w[R.inj_gen_p + i]);
      }
    }
    for (int i = q + EPT * TPP; i < R.n_load; i += TPP) {
      const int b = inj_bus[R.inj_load_p + i];
      if (b >= 0) atomicAdd(&Ar[b], (double)(ch[i] * (sc ? sc[i] : 1.f)) * P.inj_w[R.inj_load_p + i]);
    }
    for (int i = q + EPT * TPP; i < R.n_gen; i += TPP) {
      const int b = inj_bus[R.inj_gen_p + i];
      if (b < 0) continue;
      float pp = ch[2 * R.n_load + i] * sp;
      if (R.lane_gen_delta) pp += R.lane_gen_delta[(size_t)lane * R.n_gen + i];
      atomicAdd(&Ar[b], (double)pp * P.inj_w[R.inj_gen_p + i]);
    }
    const double* irow = inj + (size_t)lane * P.n_inj;
    for (int i = R.inj_sto_p + q; i < R.inj_sto_p + R.n_inj_tail; i += TPP) {
      const int b = inj_bus[i];
      if (b >= 0) atomicAdd(&Ar[b], irow[i] * P.inj_w[i]);
    }
  }
  __syncthreads();
  // ---- GEMM: wavefront w owns the line tiles w, w + 4, ...; MT row tiles share every B operand; the operands of trip s + 1 are
  //      fetched (L2 / LDS) before the 4 MT MFMAs of trip s issue --------------------------------------------------------------------
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int KT = 8;                                     // k-steps per trip
  const int trips = R.kpad / (4 * KT);
  const int n_tiles = P.line_pad / 16;
  const double* arow = A + (size_t)(l & 15) * stride + (l >> 4);
  for (int t = w; t < n_tiles; t += 4) {
    v4d c[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) c[m] = v4d{0.0, 0.0, 0.0, 0.0};
    const double* bcol = ptdf_t + (size_t)(l >> 4) * P.line_pad + t * 16 + (l & 15);
    double a[MT][KT], b[KT];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      b[u] = bcol[(size_t)4 * u * P.line_pad];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m][u] = arow[(size_t)16 * m * stride + 4 * u];
    }
    for (int s = 0; s < trips; ++s) {
      double an[MT][KT], bn[KT];
      const int sn = s + 1 < trips ? s + 1 : s;             // (the last trip re-fetches its own operands: no branch in the loop)
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        bn[u] = bcol[(size_t)4 * (KT * sn + u) * P.line_pad];
#pragma unroll
        for (int m = 0; m < MT; ++m) an[m][u] = arow[(size_t)16 * m * stride + 4 * (KT * sn + u)];
      }
#pragma unroll
      for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int m = 0; m < MT; ++m) c[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][u], b[u], c[m], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        b[u] = bn[u];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m][u] = an[m][u];
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int pi = 16 * m + 4 * v + (l >> 4);
        const int rw = p_row[pi], ln = p_lane[pi];
        if (rw >= 0) flow[((size_t)rw * lane_stride + ln) * P.line_pad + t * 16 + (l & 15)] = nan_out ? __builtin_nanf("") : (float)c[m][v];
      }
  }
}

// K_L: DC N-1 screening.  (The LODF table is float32 -- the screening result is float32 and the flows it multiplies are; the builder of
// per-lane tables streams one table per topology class to HBM, half the bytes.)  Post-outage flows are f_l + LODF[l][k] * f_k (rank-1 update of the pre-outage flows), so for
// every lane and every single-line outage k the worst loading max_l |f_l + LODF[l][k] f_k| * inv_cap[l] needs no solve.
// A block of 4 wavefronts serves LODF_LPW lanes; a thread owns the outages k = tid, tid + 64, ... and each wavefront walks a
// QUARTER of the monitored lines l (8 LODF loads in flight per thread; every element fetched from L2 is used for all the lanes of
// the block); the four partial maxima are combined through LDS.
constexpr int LODF_LPW = 4;
__global__ __launch_bounds__(256) void lodf_screen_kernel(int n_line, int line_pad, const float* __restrict__ lodf /* [n_line][line_pad], float32 */,
                                                           const float* __restrict__ inv_cap /* [n_line] or nullptr */,
                                                           const float* __restrict__ flow, int lane0, int n_lanes,
                                                           float* __restrict__ worst /* [lanes][line_pad], indexed by lane */,
                                                           const int* __restrict__ order = nullptr /* per-lane topologies: slot -> lane (PtdfDev::order) */,
                                                           const int* __restrict__ blk_class = nullptr, long long lodf_stride = 0,
                                                           const int* __restrict__ cls_status = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* f = reinterpret_cast<float*>(smem);                     // [LODF_LPW][line_pad]
  float* ic = f + (size_t)LODF_LPW * line_pad;                   // [line_pad]
  float* part = ic + line_pad;                                   // [4 wavefronts][LODF_LPW][line_pad] partial maxima (+inf: islanding)
  const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6, r0 = blockIdx.x * LODF_LPW;
  __shared__ int s_ln[LODF_LPW];
  bool nan_out = false;
  if (order) {                                                   // the LODF_LPW slots of a block lie in one group of 16: one class
    const int cls = blk_class[r0 / 16];
    lodf += (size_t)cls * lodf_stride;
    nan_out = cls_status[cls] != 0;
    if (threadIdx.x < LODF_LPW) s_ln[threadIdx.x] = order[r0 + threadIdx.x];
  } else if (threadIdx.x < LODF_LPW) s_ln[threadIdx.x] = (r0 + (int)threadIdx.x < n_lanes) ? lane0 + r0 + (int)threadIdx.x : -1;
  __syncthreads();
  for (int i = threadIdx.x; i < LODF_LPW * line_pad; i += 256) {
    const int ln = s_ln[i / line_pad], l = i % line_pad;
    f[i] = (ln >= 0 && l < n_line) ? flow[(size_t)ln * line_pad + l] : 0.f;
  }
  for (int l = threadIdx.x; l < line_pad; l += 256) ic[l] = l < n_line ? (inv_cap ? inv_cap[l] : 1.f) : 0.f;
  __syncthreads();
  const int q = (n_line + 3) / 4, l_beg = wv * q, l_end = (l_beg + q < n_line) ? l_beg + q : n_line;
  for (int k = tid; k < n_line; k += WAVE) {
    double fk[LODF_LPW], m[LODF_LPW];
#pragma unroll
    for (int r = 0; r < LODF_LPW; ++r) { fk[r] = (double)f[r * line_pad + k]; m[r] = 0.0; }
    bool island = false;
    int l = l_beg;
    for (; l + 8 <= l_end; l += 8) {
      double d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = (double)lodf[(size_t)(l + u) * line_pad + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        island |= (d[u] != d[u]);
        const double w = (double)ic[l + u];
#pragma unroll
        for (int r = 0; r < LODF_LPW; ++r) m[r] = fmax(m[r], fabs(fma(d[u], fk[r], (double)f[r * line_pad + l + u])) * w);
      }
    }
    for (; l < l_end; ++l) {
      const double d = (double)lodf[(size_t)l * line_pad + k];
      island |= (d != d);
      const double w = (double)ic[l];
#pragma unroll
      for (int r = 0; r < LODF_LPW; ++r) m[r] = fmax(m[r], fabs(fma(d, fk[r], (double)f[r * line_pad + l])) * w);
    }
#pragma unroll
    for (int r = 0; r < LODF_LPW; ++r) part[((size_t)wv * LODF_LPW + r) * line_pad + k] = island ? __builtin_inff() : (float)m[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LODF_LPW * n_line; i += 256) {
    const int r = i / n_line, k = i % n_line;
    if (s_ln[r] < 0) continue;
    float m = part[(size_t)r * line_pad + k];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) m = fmaxf(m, part[((size_t)w2 * LODF_LPW + r) * line_pad + k]);
    worst[(size_t)s_ln[r] * line_pad + k] = nan_out ? __builtin_nanf("") : m;
  }
}

}  // namespace gpf
