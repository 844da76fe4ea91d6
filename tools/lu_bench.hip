// Developer tool: micro-benchmark of the in-LDS block-sparse LU of kernel S (gridpf_sparse.hpp: block_lu_flat) on the
// substation graph of a grid, one wavefront per block, REPS solves per launch, cycle counts per solve.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lu_bench.hip -o tools/_build/lu_bench
//   tools/_build/lu_bench graph.txt [blocks=4096] [reps=50]        (graph.txt: "n_sub n_line" then "or ex" per line)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../grid2op_amd/csrc/gridpf_common.hpp"
#include "../grid2op_amd/csrc/gridpf_sparse.hpp"
#include "../grid2op_amd/csrc/gridpf_symbolic.hpp"

using namespace gpf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int BS, int IPW>
__global__ __launch_bounds__(64, 4) void lu_kernel(SymDev S, FlatDev F, const int* __restrict__ fprog, int rslot0, const double* __restrict__ A0,
                                                    const double* __restrict__ b0, int reps, int do_solve, long long* cycles, double* xout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int GW = 64 / IPW;
  const int tid = threadIdx.x, grp = tid / GW, t = tid % GW;
  constexpr int B2 = BS * BS;
  const size_t NS = (size_t)rslot0 + S.n, HS = NS * 2;
  double* Ap = reinterpret_cast<double*>(smem);                      // pristine copy (shared by the groups)
  double* bp = Ap + (size_t)S.nslot * B2;
  double* A = bp + (size_t)S.n * BS + (size_t)grp * (NS * B2);
  int* prog = reinterpret_cast<int*>(bp + (size_t)S.n * BS + (size_t)IPW * (NS * B2));
  for (int i = tid; i < S.nslot * B2; i += 64) Ap[i] = A0[i];
  for (int i = tid; i < S.n * BS; i += 64) bp[i] = b0[i];
  for (int i = tid; i < F.n_words; i += 64) prog[i] = fprog[i];
  __syncthreads();
  bool ok = true;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    // 2x2 blocks are stored split by row: row 0 of every (pseudo-)slot first, row 1 behind; right-hand side in the pseudo-slots
    for (int i = t; i < S.nslot * B2; i += GW) { const int slot = i / B2, e = i % B2; A[(e / BS) * HS + slot * BS + (e % BS)] = Ap[i]; }
    for (int i = t; i < S.n * BS; i += GW) A[(i % BS) * HS + ((size_t)rslot0 + i / BS) * 2] = bp[i];
    __syncthreads();
    if (do_solve) ok &= block_lu_flat<GW>(F, prog, A, HS, t);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cycles[blockIdx.x] = (t1 - t0) + (ok ? 0 : 1000000000000LL);
  if (blockIdx.x == 0 && grp == IPW - 1) for (int i = t; i < S.n * BS; i += GW) xout[i] = A[(i % BS) * HS + ((size_t)rslot0 + i / BS) * 2];
}

template <int BS, int IPW>
void run(const SymDev& D, const Symbolic& S, int n, int instances, int reps, const double* dA, const double* db, double* dx, long long* dcy,
         double* med) {
  FlatProg FP = build_flat(S, 64 / IPW);
  FlatDev F{};
  F.n_fwd = FP.n_fwd; F.n_scale = FP.n_scale; F.n_scale_rhs = FP.n_scale_rhs; F.n_back = FP.n_back; F.scale_off = FP.scale_off;
  F.back_off = FP.back_off; F.rhs_field0 = FP.rhs_field0; F.n_words = (int)FP.words.size();
  int* dfp; CK(hipMalloc(&dfp, FP.words.size() * 4)); CK(hipMemcpy(dfp, FP.words.data(), FP.words.size() * 4, hipMemcpyHostToDevice));
  printf("flat program GW=%d: %d forward + %d back passes, %d scale passes, %zu ints\n", 64 / IPW, F.n_fwd, F.n_back, F.n_scale, FP.words.size());
  constexpr int B2 = BS * BS;
  const int blocks = instances / IPW;
  const size_t lds = ((size_t)S.nslot * B2 + (size_t)n * BS + (size_t)IPW * ((size_t)S.rslot0 + n) * B2) * 8 + FP.words.size() * 4 + 16;
  CK(hipFuncSetAttribute((const void*)lu_kernel<BS, IPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  printf("IPW=%d: LDS %zu B/block, blocks=%d\n", IPW, lds, blocks);
  for (int solve = 0; solve < 2; ++solve) {
    for (int w = 0; w < 2; ++w) {
      hipLaunchKernelGGL((lu_kernel<BS, IPW>), dim3(blocks), dim3(64), lds, 0, D, F, dfp, S.rslot0, dA, db, reps, solve, dcy, dx);
      CK(hipDeviceSynchronize());
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((lu_kernel<BS, IPW>), dim3(blocks), dim3(64), lds, 0, D, F, dfp, S.rslot0, dA, db, reps, solve, dcy, dx);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cy(blocks);
    CK(hipMemcpy(cy.data(), dcy, (size_t)blocks * 8, hipMemcpyDeviceToHost));
    std::sort(cy.begin(), cy.end());
    med[solve] = (double)cy[blocks / 2] / reps;
    med[2 + solve] = ms * 1000.0 / reps;
  }
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: lu_bench graph.txt [blocks] [reps]\n"); return 1; }
  FILE* f = fopen(argv[1], "r");
  if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
  int n, nl;
  if (fscanf(f, "%d %d", &n, &nl) != 2) return 1;
  std::vector<int> lo(nl), le(nl);
  for (int l = 0; l < nl; ++l) if (fscanf(f, "%d %d", &lo[l], &le[l]) != 2) return 1;
  fclose(f);
  const int blocks = argc > 2 ? atoi(argv[2]) : 4096, reps = argc > 3 ? atoi(argv[3]) : 50;
  constexpr int BS = 2, B2 = 4;
  Symbolic S = build_symbolic(n, nl, lo.data(), le.data());
  // diagonally dominant random blocks on the original pattern
  std::vector<double> A((size_t)S.nslot * B2, 0.0), b((size_t)n * BS);
  srand(1);
  auto rnd = []() { return (double)rand() / RAND_MAX - 0.5; };
  for (int s = 0; s < S.nslot_y; ++s)
    for (int m = 0; m < B2; ++m) A[(size_t)s * B2 + m] = rnd() + ((s < n && (m == 0 || m == 3)) ? 8.0 : 0.0);
  for (auto& v : b) v = rnd();
  SymDev D{};
  D.n = n; D.nslot = S.nslot; D.nslot_y = S.nslot_y; D.n_levels = S.n_levels; D.back_off = S.back_off; D.n_prog = (int)S.prog.size();
  D.scale_off = S.scale_off; D.n_scale = S.n_scale; D.back_first = S.back_first;
  int* dprog; double *dA, *db, *dx; long long* dcy;
  CK(hipMalloc(&dprog, S.prog.size() * 4)); CK(hipMemcpy(dprog, S.prog.data(), S.prog.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dA, A.size() * 8)); CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&db, b.size() * 8)); CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dx, b.size() * 8)); CK(hipMalloc(&dcy, (size_t)blocks * 8));
  D.prog = dprog;
  const int ipw = argc > 4 ? atoi(argv[4]) : 1;
  printf("n=%d nslot=%d levels=%d prog=%zu ints, instances=%d reps=%d\n", n, S.nslot, S.n_levels, S.prog.size(), blocks, reps);
  double med[4];
  if (ipw == 1) run<BS, 1>(D, S, n, blocks, reps, dA, db, dx, dcy, med);
  else if (ipw == 2) run<BS, 2>(D, S, n, blocks, reps, dA, db, dx, dcy, med);
  else run<BS, 4>(D, S, n, blocks, reps, dA, db, dx, dcy, med);
  // check the solution of block 0 against a dense solve
  std::vector<double> x(b.size());
  CK(hipMemcpy(x.data(), dx, b.size() * 8, hipMemcpyDeviceToHost));
  const int N = n * BS;
  std::vector<double> M((size_t)N * (N + 1), 0.0);
  for (int s = 0; s < S.nslot_y; ++s)
    for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) M[(size_t)(S.slot_row[s] * BS + r) * (N + 1) + S.slot_col[s] * BS + q] = A[(size_t)s * B2 + r * BS + q];
  for (int i = 0; i < N; ++i) M[(size_t)i * (N + 1) + N] = b[i];
  for (int k = 0; k < N; ++k) {
    int p = k;
    for (int r = k + 1; r < N; ++r) if (fabs(M[(size_t)r * (N + 1) + k]) > fabs(M[(size_t)p * (N + 1) + k])) p = r;
    for (int q = 0; q <= N; ++q) std::swap(M[(size_t)k * (N + 1) + q], M[(size_t)p * (N + 1) + q]);
    for (int r = 0; r < N; ++r) if (r != k) { const double m = M[(size_t)r * (N + 1) + k] / M[(size_t)k * (N + 1) + k]; for (int q = k; q <= N; ++q) M[(size_t)r * (N + 1) + q] -= m * M[(size_t)k * (N + 1) + q]; }
  }
  double err = 0;
  for (int i = 0; i < N; ++i) err = fmax(err, fabs(x[i] - M[(size_t)i * (N + 1) + N] / M[(size_t)i * (N + 1) + i]));
  printf("cycles per rep: restore-only %.0f, restore+solve %.0f  => LU solve %.0f cycles; launch us per rep: %.3f -> %.3f => %.3f us per batched solve  (max |x - dense| = %.2e)\n",
         med[0], med[1], med[1] - med[0], med[2], med[3], med[3] - med[2], err);
  return 0;
}
