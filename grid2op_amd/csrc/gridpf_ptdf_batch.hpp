// gridpf_ptdf_batch.hpp -- DC sensitivities of MANY topologies at once: one workgroup per distinct topology ("class") builds the reduced
// DC matrix B' of its topology, inverts it with a blocked Gauss-Jordan whose panel / trailing updates run on the FP64 matrix cores
// (v_mfma_f64_16x16x4_f64), and forms PTDF^T and the LODF table on the device (gpf_ptdf_build_batch).
//
// Reference: every DC power flow of the reference factorises B' of whatever topology the environment has at that moment
// (pp.rundcpp, grid2op/Backend/pandaPowerBackend.py:1090; N1Reward does it once per contingency, grid2op/Reward/n1Reward.py:70-99).
// gpf_ptdf_build (round 3) inverted ONE lane's topology on the host; a batch of 2 048 lanes with a few hundred distinct topologies
// (bus splits, outages) had no fast DC path.  Here the host only does the integer work per class (which buses are live, the compact
// bus numbering, the connectivity check of rundcpp(check_connectivity=True)); all floating point is on the device.
//
// Numbering of a class: COMPACT bus index = [active non-reference buses 0 .. nr-1 | active reference buses nr .. n_act-1].  The
// reduced matrix is over the first nr; rows / columns nr .. n_pad-1 (n_pad = nr rounded up to 16) are identity.  B' of a connected grid
// with positive branch susceptances is symmetric positive definite: every Schur complement is too, so the elimination needs no
// pivoting across tiles; a pivot that is not > 1e-12 (the host routine's test) marks the class GPF_PTDF_SINGULAR.
#pragma once
#include "gridpf_ptdf.hpp"

namespace gpf {

constexpr int PTDFB_TILE = 16;
constexpr int PTDFB_MAX_N = 256;          // reduced dimension (padded) a workgroup handles: panels of 2 x 34 KB in LDS
constexpr int PTDFB_THREADS = 256;
constexpr int PTDFB_ROWS = 4;             // rows of X a wavefront has in flight while it forms PTDF^T / LODF rows

// class descriptor (ints): [0] nr, [1] n_act, [2] n_pad, [3] host status (0 ok, 2 islanded, 3 no slack), then lf[n_line], lt[n_line]
// (compact bus of each line end; -1: line out of service or both ends on the same bus), then inj_bus[n_inj] (compact bus of each
// injection column, -1: not an active-power injection of this topology), then lflag[n_line] (1: one end of the line is a bus that
// carries NOTHING but this line end -- its outage removes that bus instead of islanding it: the reference's DC power flow of the
// contingency converges with every other flow unchanged, so the LODF column is 0 instead of NaN)
constexpr int PTDFB_HDR = 4;

struct PtdfBuildDev {
  int n_line, line_pad, n_inj, kpad;        // kpad: rows of every class's PTDF^T block (max n_act rounded up to 32)
  int desc_stride;                          // ints per class descriptor
  long long work_stride, ptdf_stride, lodf_stride;   // doubles per class
  const int* desc;                          // [n_classes][desc_stride]
  const double* br_bdc;                     // [n_line]
  long long* dbg;                           // developer: [n_classes][8] shader-clock stamps of the phases (nullptr: off)
  double* work;                             // [n_classes][n_pad_max^2] B' -> its inverse (row-major, leading dimension = the class's n_pad)
  double* ptdf_t;                           // [n_classes][kpad][line_pad]
  float* lodf;                              // [n_classes][n_line][line_pad] (float32: what the screening kernel reads) or nullptr
  int* status;                              // [n_classes] 0 ok, 1 singular pivot, 2 islanded, 3 no slack
};

__host__ __device__ inline int ptdfb_lcol_stride() { return PTDFB_TILE + 1; }
__host__ __device__ inline int ptdfb_r_stride(int n_pad) { return n_pad + 4; }
__host__ __device__ inline size_t ptdfb_lds_bytes(int n_pad_max, int line_pad) {
  // Gauss-Jordan: column panel [n_pad][17] + row panel [16][n_pad + 4] + diagonal tile [16][17];
  // tables: 1 / (1 - H[k][k]) [line_pad] + 4 wavefronts x PTDFB_ROWS rows [n_pad] + the line-end tables (3 x line_pad ints)
  const size_t gj = (size_t)n_pad_max * ptdfb_lcol_stride() + (size_t)PTDFB_TILE * ptdfb_r_stride(n_pad_max) + PTDFB_TILE * (PTDFB_TILE + 1);
  const size_t fin = (size_t)line_pad + 4 * (size_t)PTDFB_ROWS * n_pad_max + (3 * (size_t)line_pad + 1) / 2;
  return (gj > fin ? gj : fin) * sizeof(double);
}

// inverse of the 16 x 16 tile Pt (LDS, row stride 17), in place, by ONE wavefront: Gauss-Jordan without pivoting, lane l owns row
// l % 16, columns 4 (l / 16) .. + 3.  The LDS executes the DS operations of one wavefront in issue order: the reads of step p + 1
// see the writes of step p without a barrier; the compiler is held by the wave barriers.
__device__ __forceinline__ bool ptdfb_invert_tile(double* Pt, int l) {
  const int i = l & 15, c0 = 4 * (l >> 4);
  double t[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = Pt[i * 17 + c0 + q];
  bool ok = true;
#pragma unroll
  for (int p = 0; p < PTDFB_TILE; ++p) {
    const double piv = Pt[p * 17 + p];
    const double colp = Pt[i * 17 + p];
    double rowp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rowp[q] = Pt[p * 17 + c0 + q];
    ok = ok && (fabs(piv) > 1e-12);
    const double rp = 1.0 / piv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = c0 + q;
      if (i == p) t[q] = (j == p) ? rp : rowp[q] * rp;
      else t[q] = (j == p) ? -colp * rp : t[q] - colp * (rowp[q] * rp);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) Pt[i * 17 + c0 + q] = t[q];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  return __all(ok);
}

// B' of one class into M (row-major, leading dimension ld): zero fill by all threads, then thread r < nr owns row r and walks the lines
// in ascending order (no atomics: the same bits on every run)
__device__ inline void ptdfb_assemble(double* M, int ld, int n_pad, int nr, const int* lf, const int* lt, const double* __restrict__ br_bdc, int n_line, int tid) {
  for (int i = tid; i < n_pad * n_pad; i += PTDFB_THREADS) { const int r = i / n_pad, c = i - r * n_pad; M[(size_t)r * ld + c] = (r == c && r >= nr) ? 1.0 : 0.0; }
  __syncthreads();
  for (int r = tid; r < nr; r += PTDFB_THREADS) {
    double* row = M + (size_t)r * ld;
    double diag = 0.0;
    for (int k = 0; k < n_line; ++k) {
      const int a = lf[k], b = lt[k];
      if (a < 0 || b < 0 || a == b || (a != r && b != r)) continue;
      const double bb = br_bdc[k];
      diag += bb;
      const int o = a == r ? b : a;
      if (o < nr) row[o] -= bb;
    }
    row[r] = diag;
  }
}

#define PTDFB_STAMP(i_) do { if (D.dbg && threadIdx.x == 0) D.dbg[(size_t)blockIdx.x * 8 + (i_)] = (long long)__builtin_readcyclecounter(); } while (0)

// K_PB: one workgroup (4 wavefronts) per topology class.
__global__ __launch_bounds__(PTDFB_THREADS) void ptdf_build_kernel(PtdfBuildDev D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_bad;
  const int cls = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int* desc = D.desc + (size_t)cls * D.desc_stride;
  const int nr = desc[0], n_pad = desc[2], host_st = desc[3];
  const int* lf = desc + PTDFB_HDR;
  const int* lt = lf + D.n_line;
  double* M = D.work + (size_t)cls * D.work_stride;
  double* PT = D.ptdf_t + (size_t)cls * D.ptdf_stride;
  float* LO = D.lodf ? D.lodf + (size_t)cls * D.lodf_stride : nullptr;
  if (tid == 0) s_bad = 0;
  PTDFB_STAMP(0);
  if (host_st != 0) {                                   // islanded topology / no slack: no sensitivities (the flows kernels write NaN)
    for (int i = tid; i < D.kpad * D.line_pad; i += PTDFB_THREADS) PT[i] = 0.0;
    if (LO) for (int i = tid; i < D.n_line * D.line_pad; i += PTDFB_THREADS) LO[i] = 0.f;
    if (tid == 0) D.status[cls] = host_st;
    return;
  }
  // ---- 1. assemble B' (reduced: rows / columns of the reference buses dropped; padding = identity) ------------------------------
  //         one thread per row, lines in ascending order: no atomics, the same bits on every run
  ptdfb_assemble(M, n_pad, n_pad, nr, lf, lt, D.br_bdc, D.n_line, tid);
  __syncthreads();
  PTDFB_STAMP(1);
  // ---- 2. in-place inverse: blocked Gauss-Jordan, 16 x 16 tiles, updates on the FP64 matrix cores ---------------------------------
  //   step k:  P = inv(A_kk);  A_kj <- P A_kj (j != k);  A_ij <- A_ij - A_ik A_kj (i, j != k);  A_ik <- -A_ik P;  A_kk <- P
  // MFMA operand layout (gridpf_ptdf.hpp): A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16], D[i = 4 v + l / 16][j = l % 16].
  double* Lc = reinterpret_cast<double*>(smem);                       // column panel A_:,k   [n_pad][17]
  double* R = Lc + (size_t)n_pad * ptdfb_lcol_stride();               // row panel    A_k,:   [16][n_pad + 4]
  const int ldr = ptdfb_r_stride(n_pad);
  double* Pt = R + (size_t)PTDFB_TILE * ldr;                          // diagonal tile [16][17]
  const int N = n_pad / PTDFB_TILE;
  long long acc_ld = 0, acc_inv = 0, acc_tr = 0, c0_ = 0, c1_ = 0;          // developer stamps (D.dbg): panel loads / tile inversions / trailing updates
  for (int k = 0; k < N; ++k) {
    if (D.dbg) c0_ = (long long)__builtin_readcyclecounter();
    for (int i = tid; i < n_pad * PTDFB_TILE; i += PTDFB_THREADS) { const int r = i >> 4, c = i & 15; Lc[r * 17 + c] = M[(size_t)r * n_pad + k * 16 + c]; }
    for (int i = tid; i < PTDFB_TILE * n_pad; i += PTDFB_THREADS) { const int r = i / n_pad, c = i - r * n_pad; R[r * ldr + c] = M[(size_t)(k * 16 + r) * n_pad + c]; }
    __syncthreads();
    if (D.dbg) { c1_ = (long long)__builtin_readcyclecounter(); acc_ld += c1_ - c0_; }
    if (w == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int e = l + 64 * q; Pt[(e >> 4) * 17 + (e & 15)] = Lc[(k * 16 + (e >> 4)) * 17 + (e & 15)]; }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (!ptdfb_invert_tile(Pt, l) && l == 0) s_bad = 1;
    }
    __syncthreads();
    if (D.dbg) { c0_ = (long long)__builtin_readcyclecounter(); acc_inv += c0_ - c1_; }
    // row panel: R_j <- P R_j (tile k itself becomes P); written back to the matrix as well
    for (int j = w; j < N; j += 4) {
      v4d c = {0.0, 0.0, 0.0, 0.0};
      if (j != k) {
        double a[4], b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { a[s] = Pt[(l & 15) * 17 + 4 * s + (l >> 4)]; b[s] = R[(4 * s + (l >> 4)) * ldr + j * 16 + (l & 15)]; }
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], c, 0, 0, 0);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) c[v] = Pt[(4 * v + (l >> 4)) * 17 + (l & 15)];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * v + (l >> 4), cc = j * 16 + (l & 15);
        R[r * ldr + cc] = c[v];
        M[(size_t)(k * 16 + r) * n_pad + cc] = c[v];
      }
    }
    __syncthreads();
    if (D.dbg) c1_ = (long long)__builtin_readcyclecounter();
    // trailing update + column panel: tile (i, j), i != k:  C <- (j == k ? 0 : C) - L_i R_j   (R_k holds P)
    const int n_t = (N - 1) * N;
    for (int q = w; q < n_t; q += 4) {
      int i = q / N;
      const int j = q - i * N;
      if (i >= k) ++i;
      double a[4], b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { a[s] = -Lc[(i * 16 + (l & 15)) * 17 + 4 * s + (l >> 4)]; b[s] = R[(4 * s + (l >> 4)) * ldr + j * 16 + (l & 15)]; }
      double* Ct = M + (size_t)(i * 16 + (l >> 4)) * n_pad + j * 16 + (l & 15);
      v4d c = {0.0, 0.0, 0.0, 0.0};
      if (j != k) {
#pragma unroll
        for (int v = 0; v < 4; ++v) c[v] = Ct[(size_t)4 * v * n_pad];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], c, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; ++v) Ct[(size_t)4 * v * n_pad] = c[v];
    }
    __syncthreads();
    if (D.dbg) acc_tr += (long long)__builtin_readcyclecounter() - c1_;
  }
  if (D.dbg && tid == 0) { D.dbg[(size_t)cls * 8 + 2] = acc_inv; D.dbg[(size_t)cls * 8 + 6] = acc_tr; D.dbg[(size_t)cls * 8 + 7] = acc_ld; }
  const bool bad = s_bad != 0;
  PTDFB_STAMP(3);
  // ---- 3. PTDF^T[b][k] = bdc_k (X[b][from_k] - X[b][to_k])  (X symmetric; reference buses: zero rows and zero terms) ---------------
  // A wavefront takes PTDFB_ROWS rows b of X at a time into LDS (coalesced, all loads in flight together), then its lanes walk the lines.
  double* hden = reinterpret_cast<double*>(smem);                     // [line_pad] 1 / (1 - H[k][k]); 0: column of zeros; NaN: islanding outage
  double* rows = hden + D.line_pad;                                   // [4 wavefronts][PTDFB_ROWS][n_pad]
  int* s_lf = reinterpret_cast<int*>(rows + (size_t)4 * PTDFB_ROWS * n_pad);   // line-end tables of the class
  int* s_lt = s_lf + D.line_pad;
  int* s_fl = s_lt + D.line_pad;
  const int* lflag = desc + PTDFB_HDR + 2 * D.n_line + D.n_inj;
  for (int k = tid; k < D.line_pad; k += PTDFB_THREADS) {
    const bool in = k < D.n_line;
    const int f = in ? lf[k] : -1, t = in ? lt[k] : -1;
    const bool on = f >= 0 && t >= 0 && f != t;
    s_lf[k] = on ? f : -1; s_lt[k] = on ? t : -1; s_fl[k] = in ? lflag[k] : 0;
  }
  __syncthreads();
  double* wrow = rows + (size_t)w * PTDFB_ROWS * n_pad;
  for (int b0 = w * PTDFB_ROWS; b0 < D.kpad; b0 += 4 * PTDFB_ROWS) {
#pragma unroll
    for (int u = 0; u < PTDFB_ROWS; ++u) {
      const int b = b0 + u;
      for (int c = l; c < n_pad; c += 64) wrow[u * n_pad + c] = (!bad && b < nr) ? M[(size_t)b * n_pad + c] : 0.0;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < PTDFB_ROWS; ++u) {
      const int b = b0 + u;
      if (b >= D.kpad) break;
      for (int k = l; k < D.line_pad; k += 64) {
        const int f = s_lf[k], t = s_lt[k];
        double v = 0.0;
        if (f >= 0) v = D.br_bdc[k] * ((f < nr ? wrow[u * n_pad + f] : 0.0) - (t < nr ? wrow[u * n_pad + t] : 0.0));
        PT[(size_t)b * D.line_pad + k] = v;
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (tid == 0) D.status[cls] = bad ? 1 : 0;
  PTDFB_STAMP(4);
  if (!LO) return;
  __syncthreads();
  // ---- 4. LODF[m][k] = H[m][k] / (1 - H[k][k]),  H[m][k] = PTDF[m][from_k] - PTDF[m][to_k];  LODF[k][k] = -1;  NaN column: the outage
  //         of k islands the grid (as gpf_ptdf_build); column of zeros: line k is out of service, or its outage only removes a bus that
  //         carries nothing else (lflag) ---------------------------------------------------------------------------------------------
  for (int k = tid; k < D.line_pad; k += PTDFB_THREADS) {
    double d = 0.0;
    const int f = s_lf[k], t = s_lt[k];
    if (f >= 0) {
      const double den = 1.0 - (PT[(size_t)f * D.line_pad + k] - PT[(size_t)t * D.line_pad + k]);   // (rows >= nr of PT are zero)
      d = fabs(den) < 1e-8 ? (s_fl[k] ? 0.0 : __builtin_nan("")) : 1.0 / den;
    }
    hden[k] = d;
  }
  __syncthreads();
  for (int m0 = w * PTDFB_ROWS; m0 < D.n_line; m0 += 4 * PTDFB_ROWS) {
#pragma unroll
    for (int u = 0; u < PTDFB_ROWS; ++u) {                              // PTDF rows of the lines m0 .. m0 + 3 over the reduced buses (coalesced rows of X)
      const int m = m0 + u;
      const int fm = m < D.n_line ? s_lf[m] : -1, tm = m < D.n_line ? s_lt[m] : -1;
      const bool on_m = !bad && fm >= 0;
      const double bm = on_m ? D.br_bdc[m] : 0.0;
      for (int b = l; b < n_pad; b += 64)
        wrow[u * n_pad + b] = (on_m && b < nr) ? bm * ((fm < nr ? M[(size_t)fm * n_pad + b] : 0.0) - (tm < nr ? M[(size_t)tm * n_pad + b] : 0.0)) : 0.0;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < PTDFB_ROWS; ++u) {
      const int m = m0 + u;
      if (m >= D.n_line) break;
      for (int k = l; k < D.line_pad; k += 64) {
        const int f = s_lf[k], t = s_lt[k];
        double v = 0.0;
        if (f >= 0) {                                                  // (an open line: its outage changes nothing -> column of zeros)
          const double hd = hden[k];
          const double h = (f < nr ? wrow[u * n_pad + f] : 0.0) - (t < nr ? wrow[u * n_pad + t] : 0.0);
          v = (hd != hd) ? hd : (m == k ? -1.0 : h * hd);
        }
        LO[(size_t)m * D.line_pad + k] = (float)v;
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  PTDFB_STAMP(5);
}



// K_PB, reduced dimension <= 128 (the 118-substation grids: 117 .. 128 non-reference buses): REGISTER-RESIDENT blocked Gauss-Jordan.
// Wavefront w owns block row w: its (up to) 8 tiles stay in its registers (32 accumulator doubles per lane) for the whole elimination, so
// a trailing update is 4 operand reads + 4 MFMAs per tile -- no result tile goes through LDS.  What makes that possible:
//   * the MFMA result layout D[i = 4 v + l / 16][j = l % 16] IS the B-operand layout B[k = 4 s + l / 16][j = l % 16]: a tile in registers
//     is a B operand as it stands;
//   * the matrix is symmetric and the partially inverted matrix of an in-place Gauss-Jordan keeps A_ik = +-(A_ki)^T (minus for an already
//     eliminated block row i < k): the A operand of L_i = A_ik, element (l % 16, 4 s + l / 16), is element (4 s + l / 16, l % 16) of tile
//     A_ki of block row k -- the natural layout of the row the pivot wavefront publishes.  No transposition anywhere.
// Step k: wavefront k publishes its row (old values) to LDS; every wavefront j forms R_j = P A_kj (tile k: P) from it and publishes the new
// row; wavefront k reloads its row, the others update their 8 tiles; then wavefront k + 1 inverts its tile (k + 1, k + 1) -- a chain of 8
// dependent 2 x 2 pivot blocks -- ALONE: the FP64 vector unit its pivots run on is the one the other wavefronts' FP64 MFMAs occupy (64 cycles
// each), so an inversion that overlaps with a trailing update (a ninth "inverter" wavefront, or lookahead inside the update: both built
// and measured) takes 7.3 k cycles per tile instead of the ~3 k it takes with the SIMDs to itself.
// The matrix goes through LDS twice: assembled there (one thread per row, sorted line lists: deterministic), and stored back as the inverse
// for the table phases, which write PTDF^T and LODF one COLUMN per thread (coalesced stores, every operand from LDS).
constexpr int PTDFB_PT = 18;                 // row stride (doubles) of a diagonal-tile buffer: 16-byte aligned quads, conflict-free columns
constexpr int PTDFB_NT = 8;                  // block rows = worker wavefronts (reduced dimension <= 128)
constexpr int PTDFB_LDS_THREADS = 64 * PTDFB_NT;         // one wavefront per block row
constexpr int PTDFB_LROWS = 8;               // PTDF rows per round of the LODF phase
__host__ __device__ inline int ptdfb_ldm(int n_pad) { return n_pad + 2; }
__host__ __device__ inline size_t ptdfb_lds_bytes_resident(int n_pad_max, int line_pad, int n_line) {
  // matrix [n_pad][n_pad + 2] (during the elimination: the two row panels) | 1 / (1 - H[k][k]) [line_pad] + LODF round rows
  // [PTDFB_LROWS][n_pad] | two diagonal tiles [16][18] | br_bdc [line_pad] | lf, lt, lflag [line_pad] ints | row pointers + entries of B'
  const size_t dbl = (size_t)n_pad_max * ptdfb_ldm(n_pad_max) + (size_t)line_pad + (size_t)PTDFB_LROWS * n_pad_max + 2 * PTDFB_TILE * PTDFB_PT + line_pad;
  const size_t ints = 3 * (size_t)line_pad + (size_t)n_pad_max + 1 + 2 * (size_t)n_line;
  return dbl * sizeof(double) + ((ints + 3) & ~(size_t)3) * sizeof(int);
}

// inverse of the 16 x 16 tile Pt (LDS, row stride PTDFB_PT), in place, by ONE wavefront: Gauss-Jordan with 2 x 2 PIVOT BLOCKS.  The chain of
// dependent pivots is what bounds the whole elimination (the tile of step k + 1 cannot be inverted before step k updated it): a scalar pivot
// step costs an LDS round trip (the tile is exchanged through LDS, ~130 cycles under load) + a reciprocal chain + the update ~ 450 cycles; a
// 2 x 2 block step does two pivots per round trip with ONE reciprocal (of the block's determinant): 8 steps instead of 16.
// Lane l owns row l % 16, columns 4 (l / 16) .. + 3; every 16-byte LDS access below is aligned (row stride 18, even pivot index).
// Pivot test as the scalar routine's (each of the two scalar pivots > 1e-12 in magnitude): |a| and |det / a|.
__device__ __forceinline__ bool ptdfb_invert_tile_fast(double* Pt, int l) {   // (forceinline: Pt must be KNOWN to be LDS at the call site -- as a real function it compiled to flat loads / stores, 4 x slower)
  const int i = l & 15, g = l >> 4, c0 = 4 * g;
  double t[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = Pt[i * PTDFB_PT + c0 + q];
  bool ok = true;
#pragma unroll
  for (int p = 0; p < PTDFB_TILE; p += 2) {
    const double a = Pt[p * PTDFB_PT + p], b = Pt[p * PTDFB_PT + p + 1], c = Pt[(p + 1) * PTDFB_PT + p], d = Pt[(p + 1) * PTDFB_PT + p + 1];
    double C0 = Pt[i * PTDFB_PT + p], C1 = Pt[i * PTDFB_PT + p + 1];
    double R0[4], R1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { R0[q] = Pt[p * PTDFB_PT + c0 + q]; R1[q] = Pt[(p + 1) * PTDFB_PT + c0 + q]; }
    const double det = fma(a, d, -b * c);
    ok = ok && (fabs(a) > 1e-12) && (fabs(det) > 1e-12 * fabs(a));
    const double rdet = fast_rcp(det);
    const double ia = d * rdet, ib = -b * rdet, ic = -c * rdet, id = a * rdet;      // inverse of the pivot block
    // ONE formula for every element, v = keep t - (C0' U0 + C1' U1), instead of a select tree per element (the VALU work of the selects,
    // not the LDS round trip, was what a pivot step cost):
    //   rows:    general (C0, C1, keep t)      pivot row p (-1, 0, drop t)       pivot row p + 1 (0, -1, drop t)
    //   columns: general U = the new pivot rows (Binv R)[., j]      column p: U = (ia, ic), drop t      column p + 1: U = (ib, id), drop t
    const bool rp0 = i == p, rp1 = i == p + 1, pg = g == p / 4;                      // (pg: this lane's column group holds the pivot columns)
    C0 = rp0 ? -1.0 : rp1 ? 0.0 : C0;
    C1 = rp0 ? 0.0 : rp1 ? -1.0 : C1;
    const bool keep_row = !(rp0 || rp1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double U0 = fma(ia, R0[q], ib * R1[q]), U1 = fma(ic, R0[q], id * R1[q]);
      bool keep = keep_row;
      if (q == p % 4) { U0 = pg ? ia : U0; U1 = pg ? ic : U1; keep = keep && !pg; }          // (q == p % 4 is a compile-time fact)
      if (q == p % 4 + 1) { U0 = pg ? ib : U0; U1 = pg ? id : U1; keep = keep && !pg; }
      t[q] = (keep ? t[q] : 0.0) - fma(C0, U0, C1 * U1);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) Pt[i * PTDFB_PT + c0 + q] = t[q];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  return __all(ok);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the global stores in flight (s_waitcnt vmcnt(0)) -- in the table
// phases that is a full HBM write round trip per round
#define PTDFB_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(PTDFB_LDS_THREADS) void ptdf_build_lds_kernel(PtdfBuildDev D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_bad;
  const int cls = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int* desc = D.desc + (size_t)cls * D.desc_stride;
  const int nr = desc[0], n_pad = desc[2], host_st = desc[3];
  double* PT = D.ptdf_t + (size_t)cls * D.ptdf_stride;
  float* LO = D.lodf ? D.lodf + (size_t)cls * D.lodf_stride : nullptr;
  if (tid == 0) s_bad = 0;
  PTDFB_STAMP(0);
  if (host_st != 0) {
    for (int i = tid; i < D.kpad * D.line_pad; i += PTDFB_LDS_THREADS) PT[i] = 0.0;
    if (LO) for (int i = tid; i < D.n_line * D.line_pad; i += PTDFB_LDS_THREADS) LO[i] = 0.f;
    if (tid == 0) D.status[cls] = host_st;
    return;
  }
  const int ldm = ptdfb_ldm(n_pad);
  double* M = reinterpret_cast<double*>(smem);                        // [n_pad][ldm]: B', later its inverse
  double* Rold = M;                                                   // during the elimination: block row k as it was [16][ldm]
  double* Rnew = M + (size_t)PTDFB_TILE * ldm;                        // ... and after the row-panel update (tile k = P)
  double* hden = M + (size_t)n_pad * ldm;                             // [line_pad] 1 / (1 - H[k][k]); 0: column of zeros; NaN: islanding outage
  double* rows = hden + D.line_pad;                                   // [PTDFB_LROWS][n_pad] (staging rows of the LODF phase until round 6: unused, the layout is kept)
  double* Pa = rows + (size_t)PTDFB_LROWS * n_pad;                    // inverse of the current diagonal tile [16][18]
  double* Pb = Pa + PTDFB_TILE * PTDFB_PT;                            // ... of the next one (lookahead)
  double* s_bdc = Pb + PTDFB_TILE * PTDFB_PT;                         // [line_pad]
  int* s_lf = reinterpret_cast<int*>(s_bdc + D.line_pad);             // [line_pad] compact bus of the line's origin (-1: line not in the DC graph)
  int* s_lt = s_lf + D.line_pad;
  int* s_fl = s_lt + D.line_pad;
  int* s_ptr = s_fl + D.line_pad;                                     // [n_pad + 1] row r of B': entries s_ent[s_ptr[r] .. s_ptr[r + 1])
  int* s_ent = s_ptr + n_pad + 1;                                     // line | other bus << 16, lines ascending
  {
    const int* lf = desc + PTDFB_HDR;
    const int* lt = lf + D.n_line;
    const int* lflag = lt + D.n_line + D.n_inj;
    const int* cptr = lflag + D.n_line;
    const int* cent = cptr + PTDFB_MAX_N + 1;
    for (int k = tid; k < D.line_pad; k += PTDFB_LDS_THREADS) {
      const bool in = k < D.n_line;
      const int f = in ? lf[k] : -1, t = in ? lt[k] : -1;
      const bool on = f >= 0 && t >= 0 && f != t;
      s_lf[k] = on ? f : -1; s_lt[k] = on ? t : -1; s_fl[k] = in ? lflag[k] : 0;
      s_bdc[k] = in ? D.br_bdc[k] : 0.0;
    }
    for (int r = tid; r <= n_pad; r += PTDFB_LDS_THREADS) s_ptr[r] = cptr[r < nr ? r : nr];
    for (int i = tid; i < 2 * D.n_line; i += PTDFB_LDS_THREADS) s_ent[i] = cent[i];
    for (int i = tid; i < n_pad * n_pad; i += PTDFB_LDS_THREADS) { const int r = i / n_pad, c = i - r * n_pad; M[(size_t)r * ldm + c] = (r == c && r >= nr) ? 1.0 : 0.0; }
  }
  __syncthreads();
  // ---- 1. B': thread r owns row r and walks its lines in ascending order (no atomics: the same bits on every run) --------------------
  for (int r = tid; r < nr; r += PTDFB_LDS_THREADS) {
    double* row = M + (size_t)r * ldm;
    double diag = 0.0;
    for (int e = s_ptr[r]; e < s_ptr[r + 1]; ++e) {
      const int w_ = s_ent[e], o = w_ >> 16;
      const double bb = s_bdc[w_ & 0xFFFF];
      diag += bb;
      if (o < nr) row[o] -= bb;
    }
    row[r] = diag;
  }
  __syncthreads();
  PTDFB_STAMP(1);
  // ---- 2. register-resident blocked Gauss-Jordan -------------------------------------------------------------------------------------------
  const int N = n_pad / PTDFB_TILE;
  const bool worker = w < N;                                          // owns block row w
  const int g = l >> 4, cc = l & 15;
  v4d c[PTDFB_NT];                                                    // tile (w, j) in the MFMA result layout: c[j][v] = A[16 w + 4 v + g][16 j + cc]
#pragma unroll
  for (int j = 0; j < PTDFB_NT; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) c[j][v] = (worker && j < N) ? M[(size_t)(w * 16 + 4 * v + g) * ldm + j * 16 + cc] : 0.0;
  long long acc_inv = 0, acc_tr = 0, c0_ = 0;
  __syncthreads();                                                    // the matrix left LDS: its region now carries the row panels
  if (w == 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) Pa[(4 * v + g) * PTDFB_PT + cc] = c[0][v];
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (w == 0 && !ptdfb_invert_tile_fast(Pa, l) && l == 0) s_bad = 1;
  for (int k = 0; k < N; ++k) {
    double* Pt = (k & 1) ? Pb : Pa;
    double* Pn = (k & 1) ? Pa : Pb;
    if (D.dbg) c0_ = (long long)__builtin_readcyclecounter();
    if (w == k) {                                                     // publish block row k as it is
#pragma unroll
      for (int j = 0; j < PTDFB_NT; ++j)
        if (j < N) {
#pragma unroll
          for (int v = 0; v < 4; ++v) Rold[(size_t)(4 * v + g) * ldm + j * 16 + cc] = c[j][v];
        }
    }
    __syncthreads();                                                  // (also: the inverse of tile k is complete)
    if (worker) {                                                     // row panel, one tile per wavefront: R_w = P A_kw (tile k: P itself)
      v4d r = {0.0, 0.0, 0.0, 0.0};
      if (w != k) {
        double a[4], b[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) { a[s_] = Pt[cc * PTDFB_PT + 4 * s_ + g]; b[s_] = Rold[(size_t)(4 * s_ + g) * ldm + w * 16 + cc]; }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s_], b[s_], r, 0, 0, 0);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) r[v] = Pt[(4 * v + g) * PTDFB_PT + cc];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) Rnew[(size_t)(4 * v + g) * ldm + w * 16 + cc] = r[v];
    }
    __syncthreads();
    // trailing update: block row i = w != k:  C_ij <- (j == k ? 0 : C_ij) - L_i R_j with L_i = A_ik = sigma (A_ki)^T read from the OLD row k
    double a[4];
    if (worker && w != k) {
      const double sg = w < k ? 1.0 : -1.0;                           // a = -L_i = -sigma A_ki^T, sigma = -1 for an eliminated block row
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) a[s_] = sg * Rold[(size_t)(4 * s_ + g) * ldm + w * 16 + cc];
    }
    auto upd = [&](int j) {                                           // (j is a compile-time constant at every call site: c[] stays in registers)
      double b[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) b[s_] = Rnew[(size_t)(4 * s_ + g) * ldm + j * 16 + cc];
      v4d t = c[j];
      if (j == k) t = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s_], b[s_], t, 0, 0, 0);
      c[j] = t;
    };
    const int jla = k + 1 < N ? k + 1 : -1;                           // block row whose diagonal tile is the next pivot
    if (worker && w == k) {                                           // the pivot wavefront takes its new row back
#pragma unroll
      for (int j = 0; j < PTDFB_NT; ++j)
        if (j < N) {
#pragma unroll
          for (int v = 0; v < 4; ++v) c[j][v] = Rnew[(size_t)(4 * v + g) * ldm + j * 16 + cc];
        }
    }
    if (worker && w != k) {
#pragma unroll
      for (int j = 0; j < PTDFB_NT; ++j)
        if (j < N) upd(j);
    }
    __syncthreads();                                                  // every MFMA of the step has retired: the FP64 unit is free
    if (worker && w == jla) {                                         // the next diagonal tile, inverted by its owner while the others wait
      long long t0_ = D.dbg ? (long long)__builtin_readcyclecounter() : 0;
#pragma unroll
      for (int j = 0; j < PTDFB_NT; ++j)
        if (j == jla) {
#pragma unroll
          for (int v = 0; v < 4; ++v) Pn[(4 * v + g) * PTDFB_PT + cc] = c[j][v];
        }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (!ptdfb_invert_tile_fast(Pn, l) && l == 0) s_bad = 1;
      if (D.dbg) acc_inv += (long long)__builtin_readcyclecounter() - t0_;
    }
    __syncthreads();
    if (D.dbg) acc_tr += (long long)__builtin_readcyclecounter() - c0_;
  }
  if (D.dbg && tid == 0) { D.dbg[(size_t)cls * 8 + 6] = acc_tr; D.dbg[(size_t)cls * 8 + 7] = 0; }
  if (D.dbg && l == 0 && acc_inv) atomicAdd(reinterpret_cast<unsigned long long*>(&D.dbg[(size_t)cls * 8 + 2]), (unsigned long long)acc_inv);
  // the inverse back to LDS for the table phases
  if (worker) {
#pragma unroll
    for (int j = 0; j < PTDFB_NT; ++j)
      if (j < N) {
#pragma unroll
        for (int v = 0; v < 4; ++v) M[(size_t)(w * 16 + 4 * v + g) * ldm + j * 16 + cc] = c[j][v];
      }
  }
  __syncthreads();
  const bool bad = s_bad != 0;
  PTDFB_STAMP(3);
  // ---- 3. PTDF^T[b][k] = bdc_k (X[b][from_k] - X[b][to_k]): thread k owns column k (coalesced stores, operands from LDS) ----------------
  auto X = [&](int r, int c_) -> double { return (r < nr && c_ < nr) ? M[(size_t)r * ldm + c_] : 0.0; };
  for (int k = tid; k < D.line_pad; k += PTDFB_LDS_THREADS) {
    const int f = s_lf[k], t = s_lt[k];
    const bool on = f >= 0 && !bad;
    const double bk = s_bdc[k];
    const int fc = (on && f < nr) ? f : -1, tc = (on && t < nr) ? t : -1;
#pragma unroll 8
    for (int b = 0; b < D.kpad; ++b) {
      double v = 0.0;
      if (b < nr) v = bk * ((fc >= 0 ? M[(size_t)b * ldm + fc] : 0.0) - (tc >= 0 ? M[(size_t)b * ldm + tc] : 0.0));
      PT[(size_t)b * D.line_pad + k] = v;
    }
    double d = 0.0;
    if (on) {
      const double den = 1.0 - (bk * (X(f, f) - X(f, t)) - bk * (X(t, f) - X(t, t)));      // 1 - (PTDF[k][f] - PTDF[k][t]) as stored in PT
      d = fabs(den) < 1e-8 ? (s_fl[k] ? 0.0 : __builtin_nan("")) : 1.0 / den;
    }
    hden[k] = d;
  }
  if (tid == 0) D.status[cls] = bad ? 1 : 0;
  PTDFB_STAMP(4);
  if (!LO) return;
  // ---- 4. LODF[m][k] = (PTDF[m][from_k] - PTDF[m][to_k]) / (1 - H[k][k]); LODF[k][k] = -1.  With PTDF[m][b] = bdc_m (X[from_m][b] - X[to_m][b])
  //         and X symmetric, PTDF[m][from_k] - PTDF[m][to_k] = bdc_m (PT[from_m][k] - PT[to_m][k]) / bdc_k: two entries of the PTDF^T table phase 3
  //         just wrote, on ROW from_m / to_m (the same for every thread) and the thread's own column k -- coalesced reads through L1 / L2
  //         instead of gathers of X at random columns of LDS (4 per output: the lanes' from-buses collide in the 32 eight-byte banks, 47 k
  //         cycles; before that the PTDF rows of 8 lines at a time were staged in LDS, 24 rounds x 2 barriers: 74 k of the kernel's 218 k).
  //         Thread (k, stripe) owns column k for the lines of its stripe; no barrier inside the phase. ---------------------------------------
  __syncthreads();                                                     // the PTDF^T stores of every thread have left the CU (vmcnt) before another thread reads them
  const int n_str = PTDFB_LDS_THREADS / D.line_pad;                    // stripes of lines (line_pad <= the block's threads is checked by the host)
  const int kk = tid < n_str * D.line_pad ? tid % D.line_pad : -1;
  const int str = tid / D.line_pad;
  const int kf = kk >= 0 ? s_lf[kk] : -1;
  const double khd = kk >= 0 ? hden[kk] : 0.0;
  long long acc_b = 0;
  if (kk >= 0) {
    const int per = (D.n_line + n_str - 1) / n_str;
    const int m_lo = str * per, m_hi = (m_lo + per < D.n_line) ? m_lo + per : D.n_line;
    const double kbd = s_bdc[kk];
    const double ksc = (kf >= 0 && kbd != 0.0) ? khd / kbd : 0.0;      // 1 / (bdc_k (1 - H[k][k]))
    const double* const PTk = PT + kk;
    constexpr int UN = 8;                                              // lines in flight per thread: every load of a trip is issued before the first is used
    for (int m0 = m_lo; m0 < m_hi; m0 += UN) {
      double a_[UN], b_[UN], bm_[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int m = m0 + u < m_hi ? m0 + u : m_hi - 1;
        const int fm = s_lf[m], tm = s_lt[m];
        const bool on_ = fm >= 0 && !bad;
        bm_[u] = on_ ? s_bdc[m] : 0.0;
        a_[u] = PTk[(size_t)(on_ ? fm : 0) * D.line_pad];              // (rows of reference / padding buses hold zeros)
        b_[u] = PTk[(size_t)(on_ ? tm : 0) * D.line_pad];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int m = m0 + u;
        if (m >= m_hi) break;
        double v = 0.0;
        if (kf >= 0) v = (khd != khd) ? khd : (m == kk ? -1.0 : bm_[u] * (a_[u] - b_[u]) * ksc);
        LO[(size_t)m * D.line_pad + kk] = (float)v;
      }
    }
  }
  if (D.dbg && tid == 0) D.dbg[(size_t)cls * 8 + 7] = acc_b;
  PTDFB_STAMP(5);
}

#undef PTDFB_STAMP
#undef PTDFB_LDS_BARRIER

}  // namespace gpf
