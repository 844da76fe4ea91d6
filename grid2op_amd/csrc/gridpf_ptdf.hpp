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

// K_P: bus active-power injections of each lane (MW): one wavefront per lane, LDS f64 atomics from the element lanes.
__global__ __launch_bounds__(WAVE) void ptdf_bus_injection_kernel(PtdfDev P, const double* __restrict__ inj, int lane0,
                                                                  double* __restrict__ pbus) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* acc = reinterpret_cast<double*>(smem);
  const int lane = lane0 + blockIdx.x, tid = threadIdx.x;
  for (int b = tid; b < P.nb_pad; b += WAVE) acc[b] = 0.0;
  __syncthreads();
  const double* row = inj + (size_t)lane * P.n_inj;
  for (int i = tid; i < P.n_inj; i += WAVE) {
    const int b = P.inj_bus[i];
    if (b >= 0) atomicAdd(&acc[b], row[i] * P.inj_w[i]);
  }
  __syncthreads();
  double* out = pbus + (size_t)lane * P.nb_pad;
  for (int b = tid; b < P.nb_pad; b += WAVE) out[b] = acc[b];
}

// K_G: block (x, y) of 4 wavefronts: 16 lanes x the line tiles 4y .. 4y+3 (one 16 x 16 tile per wavefront)
// MFMA operand layout (64 lanes, l = lane id): A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16],
// D[i = 4 * v + l / 16][j = l % 16] for the 4 result registers v.
__global__ __launch_bounds__(256) void ptdf_gemm_kernel(PtdfDev P, const double* __restrict__ pbus, int lane0, int n_lanes,
                                                        float* __restrict__ flow) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 16;                       // first lane (of the range) of this block
  const int ksteps = P.nb_pad / 4;
  const int my_row = row0 + (l & 15);
  const double* arow = pbus + (size_t)(lane0 + (my_row < n_lanes ? my_row : n_lanes - 1)) * P.nb_pad + (l >> 4);
  const int n_tiles = P.line_pad / 16;
  for (int t = blockIdx.y * 4 + w; t < n_tiles; t += 4 * gridDim.y) {      // one 16 x 16 tile per wavefront when gridDim.y covers the tiles
    v4d c = {0.0, 0.0, 0.0, 0.0};
    const double* bcol = P.ptdf_t + (size_t)(l >> 4) * P.line_pad + t * 16 + (l & 15);
    int s = 0;
    for (; s + 8 <= ksteps; s += 8) {                       // 8 k-steps per trip: 16 loads in flight, then 8 MFMAs
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
// One wavefront serves LODF_LPW lanes (each LODF element fetched from L2 is used for all of them); a thread owns the
// outages k = tid, tid + 64, ... and walks the monitored lines l: no cross-lane reduction at all.
constexpr int LODF_LPW = 4;
__global__ __launch_bounds__(WAVE) void lodf_screen_kernel(int n_line, int line_pad, const double* __restrict__ lodf /* [n_line][line_pad] */,
                                                           const float* __restrict__ inv_cap /* [n_line] or nullptr */,
                                                           const float* __restrict__ flow, int lane0, int n_lanes,
                                                           float* __restrict__ worst /* [n_lanes][line_pad] */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* f = reinterpret_cast<float*>(smem);                     // [LODF_LPW][line_pad]
  float* ic = f + (size_t)LODF_LPW * line_pad;                   // [line_pad]
  const int tid = threadIdx.x, r0 = blockIdx.x * LODF_LPW;
  for (int i = tid; i < LODF_LPW * line_pad; i += WAVE) {
    const int r = r0 + i / line_pad, l = i % line_pad;
    f[i] = (r < n_lanes && l < n_line) ? flow[(size_t)(lane0 + r) * line_pad + l] : 0.f;
  }
  for (int l = tid; l < line_pad; l += WAVE) ic[l] = l < n_line ? (inv_cap ? inv_cap[l] : 1.f) : 0.f;
  __syncthreads();
  for (int k = tid; k < n_line; k += WAVE) {
    double fk[LODF_LPW], m[LODF_LPW];
#pragma unroll
    for (int r = 0; r < LODF_LPW; ++r) { fk[r] = (double)f[r * line_pad + k]; m[r] = 0.0; }
    bool island = false;
    for (int l = 0; l < n_line; ++l) {
      const double d = lodf[(size_t)l * line_pad + k];
      island |= (d != d);
      const double w = (double)ic[l];
#pragma unroll
      for (int r = 0; r < LODF_LPW; ++r) m[r] = fmax(m[r], fabs(fma(d, fk[r], (double)f[r * line_pad + l])) * w);
    }
#pragma unroll
    for (int r = 0; r < LODF_LPW; ++r)
      if (r0 + r < n_lanes) worst[(size_t)(r0 + r) * line_pad + k] = island ? __builtin_inff() : (float)m[r];
  }
}

}  // namespace gpf
