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
  const int row0 = blockIdx.x * 16;                        // first lane (of the range) of this block
  for (int i = threadIdx.x; i < 16 * stride; i += 256) A[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < P.n_inj; i += 256) {
    const int b = P.inj_bus[i];
    if (b < 0) continue;
    const double w = P.inj_w[i];
    const double* col = inj + (size_t)(lane0 + row0) * P.n_inj + i;
    double v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (row0 + r < n_lanes) ? col[(size_t)r * P.n_inj] : 0.0;      // 16 loads in flight
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
    const double* bcol = P.ptdf_t + (size_t)(l >> 4) * P.line_pad + t * 16 + (l & 15);
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
      const int r = row0 + 4 * v + (l >> 4);
      if (r < n_lanes) flow[(size_t)(lane0 + r) * P.line_pad + t * 16 + (l & 15)] = (float)c[v];
    }
  }
}

// K_L: DC N-1 screening.  Post-outage flows are f_l + LODF[l][k] * f_k (rank-1 update of the pre-outage flows), so for
// every lane and every single-line outage k the worst loading max_l |f_l + LODF[l][k] f_k| * inv_cap[l] needs no solve.
// A block of 4 wavefronts serves LODF_LPW lanes; a thread owns the outages k = tid, tid + 64, ... and each wavefront walks a
// QUARTER of the monitored lines l (8 LODF loads in flight per thread; every element fetched from L2 is used for all the lanes of
// the block); the four partial maxima are combined through LDS.
constexpr int LODF_LPW = 4;
__global__ __launch_bounds__(256) void lodf_screen_kernel(int n_line, int line_pad, const double* __restrict__ lodf /* [n_line][line_pad] */,
                                                           const float* __restrict__ inv_cap /* [n_line] or nullptr */,
                                                           const float* __restrict__ flow, int lane0, int n_lanes,
                                                           float* __restrict__ worst /* [n_lanes][line_pad] */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* f = reinterpret_cast<float*>(smem);                     // [LODF_LPW][line_pad]
  float* ic = f + (size_t)LODF_LPW * line_pad;                   // [line_pad]
  float* part = ic + line_pad;                                   // [4 wavefronts][LODF_LPW][line_pad] partial maxima (+inf: islanding)
  const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6, r0 = blockIdx.x * LODF_LPW;
  for (int i = threadIdx.x; i < LODF_LPW * line_pad; i += 256) {
    const int r = r0 + i / line_pad, l = i % line_pad;
    f[i] = (r < n_lanes && l < n_line) ? flow[(size_t)(lane0 + r) * line_pad + l] : 0.f;
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
      for (int u = 0; u < 8; ++u) d[u] = lodf[(size_t)(l + u) * line_pad + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        island |= (d[u] != d[u]);
        const double w = (double)ic[l + u];
#pragma unroll
        for (int r = 0; r < LODF_LPW; ++r) m[r] = fmax(m[r], fabs(fma(d[u], fk[r], (double)f[r * line_pad + l + u])) * w);
      }
    }
    for (; l < l_end; ++l) {
      const double d = lodf[(size_t)l * line_pad + k];
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
    if (r0 + r >= n_lanes) continue;
    float m = part[(size_t)r * line_pad + k];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) m = fmaxf(m, part[((size_t)w2 * LODF_LPW + r) * line_pad + k]);
    worst[(size_t)(r0 + r) * line_pad + k] = m;
  }
}

}  // namespace gpf
