// TEST INFRASTRUCTURE: host emulation of the device block-LU (gridpf_sparse.hpp: block_lu_solve) driven by the SAME
// symbolic program that the product builds (grid2op_amd/csrc/gridpf_symbolic.hpp), so that the program layout and the
// level scheduling are verified on CPU against a dense solve (tests/test_symbolic_program.py).
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "../../grid2op_amd/csrc/gridpf_symbolic.hpp"

extern "C" {

// A_dense: [n*BS][n*BS] row-major (entries outside the fill pattern are ignored), rhs: [n*BS] -> x.
// Returns the number of levels (>= 1) or a negative error code.
int sym_emul_solve(int n_sub, int n_line, const int* line_or, const int* line_ex, int BS, const double* A_dense,
                   const double* rhs_in, double* x_out, int* stats /* nslot_y, nslot, n_levels, n_prog */, int fused) {
  gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, line_or, line_ex);
  const int B2 = BS * BS, N = n_sub * BS;
  std::vector<double> A((size_t)S.nslot * B2, 0.0), rhs(rhs_in, rhs_in + N);
  for (int s = 0; s < S.nslot; ++s)
    for (int r = 0; r < BS; ++r)
      for (int q = 0; q < BS; ++q) A[(size_t)s * B2 + r * BS + q] = A_dense[(size_t)(S.slot_row[s] * BS + r) * N + S.slot_col[s] * BS + q];
  const int* prog = S.prog.data();
  auto inv = [&](const double* D, double* Di) -> bool {   // Gauss-Jordan with partial pivoting
    std::vector<double> M((size_t)BS * 2 * BS);
    for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) { M[r * 2 * BS + q] = D[r * BS + q]; M[r * 2 * BS + BS + q] = r == q; }
    for (int k = 0; k < BS; ++k) {
      int p = k;
      for (int r = k + 1; r < BS; ++r) if (std::fabs(M[r * 2 * BS + k]) > std::fabs(M[p * 2 * BS + k])) p = r;
      if (p != k) for (int q = 0; q < 2 * BS; ++q) std::swap(M[k * 2 * BS + q], M[p * 2 * BS + q]);
      const double pv = M[k * 2 * BS + k];
      if (!(std::fabs(pv) > 1e-300)) return false;
      for (int q = 0; q < 2 * BS; ++q) M[k * 2 * BS + q] /= pv;
      for (int r = 0; r < BS; ++r) if (r != k) { const double m = M[r * 2 * BS + k]; for (int q = 0; q < 2 * BS; ++q) M[r * 2 * BS + q] -= m * M[k * 2 * BS + q]; }
    }
    for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) Di[r * BS + q] = M[r * 2 * BS + BS + q];
    return true;
  };
  if (fused) {   // mirrors the BS == 2 device path: one phase per level, deferred scaling pass
    for (int lv = 0; lv < S.n_levels; ++lv) {
      const int* h = prog + 8 * lv;
      const int c_off = h[4], n_c = h[5], r_off = h[6], n_r = h[7];
      std::vector<double> dA(A.size(), 0.0), dr(rhs.size(), 0.0);      // all items of a level read the pre-level state
      std::vector<double> Di(B2), T(B2);
      for (int o = 0; o < n_c; ++o) {
        const unsigned w0 = (unsigned)prog[c_off + 2 * o], w1 = (unsigned)prog[c_off + 2 * o + 1];
        if (!inv(&A[(size_t)(w1 >> 16) * B2], Di.data())) return -1;
        const double* Al = &A[(size_t)(w0 >> 16) * B2];
        const double* Au = &A[(size_t)(w1 & 0xffffu) * B2];
        for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) { T[r * BS + q] = 0; for (int m = 0; m < BS; ++m) T[r * BS + q] += Al[r * BS + m] * Di[m * BS + q]; }
        for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) { double acc = 0; for (int m = 0; m < BS; ++m) acc += T[r * BS + m] * Au[m * BS + q]; dA[(size_t)(w0 & 0xffffu) * B2 + r * BS + q] -= acc; }
      }
      for (int o = 0; o < n_r; ++o) {
        const unsigned w0 = (unsigned)prog[r_off + 2 * o];
        const int p = prog[r_off + 2 * o + 1];
        if (!inv(&A[(size_t)p * B2], Di.data())) return -1;
        const double* Al = &A[(size_t)(w0 & 0xffffu) * B2];
        for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) { T[r * BS + q] = 0; for (int m = 0; m < BS; ++m) T[r * BS + q] += Al[r * BS + m] * Di[m * BS + q]; }
        for (int r = 0; r < BS; ++r) { double acc = 0; for (int m = 0; m < BS; ++m) acc += T[r * BS + m] * rhs[p * BS + m]; dr[(w0 >> 16) * BS + r] -= acc; }
      }
      for (size_t i = 0; i < A.size(); ++i) A[i] += dA[i];
      for (size_t i = 0; i < rhs.size(); ++i) rhs[i] += dr[i];
    }
    std::vector<double> Di(B2), T(B2);
    for (int e = 0; e < S.n_scale; ++e) {
      const unsigned w = (unsigned)prog[S.scale_off + e];
      if (!inv(&A[(size_t)(w >> 16) * B2], Di.data())) return -1;
      double* Au = &A[(size_t)(w & 0xffffu) * B2];
      for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) { T[r * BS + q] = 0; for (int m = 0; m < BS; ++m) T[r * BS + q] += Di[r * BS + m] * Au[m * BS + q]; }
      std::memcpy(Au, T.data(), sizeof(double) * B2);
    }
    for (int p = 0; p < S.n; ++p) {
      if (!inv(&A[(size_t)p * B2], Di.data())) return -1;
      std::vector<double> t(BS, 0.0);
      for (int r = 0; r < BS; ++r) for (int m = 0; m < BS; ++m) t[r] += Di[r * BS + m] * rhs[p * BS + m];
      for (int r = 0; r < BS; ++r) rhs[p * BS + r] = t[r];
    }
  } else
  for (int lv = 0; lv < S.n_levels; ++lv) {
    const int* h = prog + 8 * lv;
    const int piv_off = h[0], n_piv = h[1], b_off = h[2], n_b = h[3], c_off = h[4], n_c = h[5], r_off = h[6], n_r = h[7];
    for (int q = 0; q < n_piv; ++q) {           // (a)
      double* Ad = &A[(size_t)prog[piv_off + q] * B2];
      std::vector<double> Di(B2);
      if (!inv(Ad, Di.data())) return -1;
      std::memcpy(Ad, Di.data(), sizeof(double) * B2);
    }
    for (int e = 0; e < n_b; ++e) {             // (b) U' = Dinv * A
      const unsigned w = (unsigned)prog[b_off + e];
      const double* Di = &A[(size_t)(w >> 16) * B2];
      double* Au = &A[(size_t)(w & 0xffffu) * B2];
      std::vector<double> T(B2, 0.0);
      for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) for (int m = 0; m < BS; ++m) T[r * BS + q] += Di[r * BS + m] * Au[m * BS + q];
      std::memcpy(Au, T.data(), sizeof(double) * B2);
    }
    for (int q = 0; q < n_piv; ++q) {           //     b' = Dinv * b
      const int p = prog[piv_off + q];
      const double* Di = &A[(size_t)p * B2];
      std::vector<double> t(BS, 0.0);
      for (int r = 0; r < BS; ++r) for (int m = 0; m < BS; ++m) t[r] += Di[r * BS + m] * rhs[p * BS + m];
      for (int r = 0; r < BS; ++r) rhs[p * BS + r] = t[r];
    }
    for (int o = 0; o < n_c; ++o) {             // (c) A[dst] -= A[l] * U'[u]
      const unsigned w0 = (unsigned)prog[c_off + 2 * o];
      const int u = prog[c_off + 2 * o + 1] & 0xffff;
      const double* Al = &A[(size_t)(w0 >> 16) * B2];
      const double* Au = &A[(size_t)u * B2];
      double* Ad = &A[(size_t)(w0 & 0xffffu) * B2];
      for (int r = 0; r < BS; ++r) for (int q = 0; q < BS; ++q) { double acc = 0; for (int m = 0; m < BS; ++m) acc += Al[r * BS + m] * Au[m * BS + q]; Ad[r * BS + q] -= acc; }
    }
    for (int o = 0; o < n_r; ++o) {             //     rhs[row] -= A[l] * b'[p]
      const unsigned w0 = (unsigned)prog[r_off + 2 * o];
      const int p = prog[r_off + 2 * o + 1];
      const double* Al = &A[(size_t)(w0 & 0xffffu) * B2];
      for (int r = 0; r < BS; ++r) { double acc = 0; for (int m = 0; m < BS; ++m) acc += Al[r * BS + m] * rhs[p * BS + m]; rhs[(w0 >> 16) * BS + r] -= acc; }
    }
  }
  for (int lv = S.n_levels - 1; lv >= 0; --lv) {   // back substitution
    const int ent_off = prog[S.back_off + 2 * lv], n_ent = prog[S.back_off + 2 * lv + 1];
    std::vector<double> delta((size_t)N, 0.0);
    for (int e = 0; e < n_ent; ++e) {
      const unsigned w = (unsigned)prog[ent_off + 2 * e];
      const int p = prog[ent_off + 2 * e + 1];
      const double* Au = &A[(size_t)(w & 0xffffu) * B2];
      const double* xj = &rhs[(size_t)(w >> 16) * BS];
      for (int r = 0; r < BS; ++r) { double acc = 0; for (int m = 0; m < BS; ++m) acc += Au[r * BS + m] * xj[m]; delta[p * BS + r] -= acc; }
    }
    for (int i = 0; i < N; ++i) rhs[i] += delta[i];
  }
  std::memcpy(x_out, rhs.data(), sizeof(double) * N);
  if (stats) { stats[0] = S.nslot_y; stats[1] = S.nslot; stats[2] = S.n_levels; stats[3] = (int)S.prog.size(); }
  return S.n_levels;
}
}

// level sizes of the program: out[4*lv + {0,1,2,3}] = n_piv, n_b, n_c, n_r; returns n_levels
extern "C" int sym_level_sizes(int n_sub, int n_line, const int* line_or, const int* line_ex, int* out, int cap, int slack) {
  gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, line_or, line_ex, slack);
  out[4 * cap] = S.nslot; out[4 * cap + 1] = S.nslot_y;
  for (int lv = 0; lv < S.n_levels && lv < cap; ++lv) {
    const int* h = S.prog.data() + 8 * lv;
    out[4 * lv] = h[1]; out[4 * lv + 1] = h[3]; out[4 * lv + 2] = h[5]; out[4 * lv + 3] = h[7];
  }
  return S.n_levels;
}

// Host emulation of the FLAT-program sweeps (gridpf_sparse.hpp: block_lu_flat; gridpf_symbolic.hpp: build_flat) for one group
// width: 2x2 blocks split by row, right-hand side in the pseudo-slots behind rslot0, every pass of exactly gw items whose
// updates are applied together (the device combines them with LDS atomics).  A_dense: [2n][2n] row-major.  Returns the number
// of passes (forward + back), or a negative error (-2: a field out of range, -3: a padding word inside the valid prefix, ...).
static int solve_flat_impl(const gpf::Symbolic& S, int n_sub, int gw, const double* A_dense, const double* rhs_in, double* x_out, int* stats);
extern "C" int sym_emul_solve_flat(int n_sub, int n_line, const int* line_or, const int* line_ex, int gw, const double* A_dense,
                                   const double* rhs_in, double* x_out, int* stats /* n_fwd, levels of the Gauss-Jordan tail, n_back, n_words */) {
  gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, line_or, line_ex);
  return solve_flat_impl(S, n_sub, gw, A_dense, rhs_in, x_out, stats);
}
// the same with the slot layout search (optimize_slot_layout: blocks renumbered inside their ranges); cost[0..1]: bank-model cost
// of the layout before / after
extern "C" int sym_emul_solve_flat_opt(int n_sub, int n_line, const int* line_or, const int* line_ex, int gw, int opt_iters, const double* A_dense,
                                       const double* rhs_in, double* x_out, int* stats, long long* cost) {
  gpf::Symbolic S0 = gpf::build_symbolic(n_sub, n_line, line_or, line_ex);
  gpf::Symbolic S = gpf::optimize_slot_layout(S0, opt_iters);
  const int gwn = n_sub <= 8 ? 16 : n_sub <= 24 ? 32 : n_sub < 64 ? 64 : 128;
  if (cost) { cost[0] = gpf::flat_layout_cost(S0, gwn); cost[1] = gpf::flat_layout_cost(S, gwn); }
  if (S.nslot != S0.nslot || S.nslot_y != S0.nslot_y || S.nslot_lu != S0.nslot_lu) return -20;
  for (int q = 0; q < n_sub; ++q) if (S.slot_row[q] != q || S.slot_col[q] != q) return -21;          // diagonal slots stay the substation ids
  for (int q = n_sub; q < S.nslot_y; ++q) {                                                           // ranges keep their members
    bool found = false;
    for (int t = n_sub; t < S0.nslot_y && !found; ++t) found = S0.slot_row[t] == S.slot_row[q] && S0.slot_col[t] == S.slot_col[q];
    if (!found) return -22;
  }
  for (int l = 0; l < n_line; ++l)
    for (int k = 0; k < 4; ++k) {
      const int q = S.br_slot[4 * l + k], r = k < 2 ? line_or[l] : line_ex[l], c2 = (k == 0 || k == 2) ? line_or[l] : line_ex[l];
      if (S.slot_row[q] != r || S.slot_col[q] != c2) return -23;
    }
  return solve_flat_impl(S, n_sub, gw, A_dense, rhs_in, x_out, stats);
}
static int solve_flat_impl(const gpf::Symbolic& S, int n_sub, int gw, const double* A_dense, const double* rhs_in, double* x_out, int* stats) {
  if (!gpf::flat_fits(S)) return -10;
  const gpf::FlatProg F = gpf::build_flat(S, gw, 400);      // with the bank-conflict-aware lane assignment
  const int N = n_sub * 2;
  const size_t NS = (size_t)S.rslot0 + n_sub, HS = NS * 2;
  std::vector<double> A(2 * HS, 0.0);                                     // row 0 of every (pseudo-)slot, then row 1
  for (int s = 0; s < S.nslot; ++s)
    for (int r = 0; r < 2; ++r)
      for (int q = 0; q < 2; ++q) A[r * HS + (size_t)s * 2 + q] = A_dense[(size_t)(S.slot_row[s] * 2 + r) * N + S.slot_col[s] * 2 + q];
  for (int p = 0; p < n_sub; ++p)
    for (int r = 0; r < 2; ++r) { A[r * HS + ((size_t)S.rslot0 + p) * 2] = rhs_in[p * 2 + r]; A[r * HS + ((size_t)S.rslot0 + p) * 2 + 1] = 12345.678; }   // pad column: garbage on purpose
  const unsigned INV = 0xffffffffu;
  auto ok_field = [&](unsigned f) { return f % 16 == 0 && f / 16 < NS; };
  auto el = [&](unsigned f, int r, int q) -> double& { return A[r * HS + (size_t)(f / 16) * 2 + q]; };
  const int* W = F.words.data();
  if (F.scale_off != 2 * gw * (F.n_fwd + 1)) return -4;                   // forward section + its padding pass
  if (!F.wave_closed) return -11;
  for (int k = 0; k < F.n_fwd; ++k) {
    std::vector<double> d(A.size(), 0.0);
    bool seen_inv = false;
    std::map<unsigned, int> dst_half;
    for (int t = 0; t < gw; ++t) {
      const unsigned w0 = (unsigned)W[2 * (k * gw + t)], w1 = (unsigned)W[2 * (k * gw + t) + 1];
      if (w0 == INV) { seen_inv = true; continue; }
      (void)seen_inv;                                       // (padding may sit anywhere in a pass: balanced splitting, wave-closed packing, lane assignment)
      const unsigned fd = w0 & 0xffffu, fl = w0 >> 16, fu = w1 & 0xffffu, fp = w1 >> 16;
      if (gw > 64) {                                        // wave-closed: every destination of a pass belongs to ONE 64-lane half
        auto it = dst_half.find(fd);
        if (it != dst_half.end() && it->second != t / 64) return -9;
        dst_half[fd] = t / 64;
      }
      if (!ok_field(fd) || !ok_field(fl) || !ok_field(fu) || !ok_field(fp)) return -2;
      const double d00 = el(fp, 0, 0), d01 = el(fp, 0, 1), d10 = el(fp, 1, 0), d11 = el(fp, 1, 1);
      const double det = d00 * d11 - d01 * d10;
      if (!(std::fabs(det) > 1e-300)) return -1;
      const double i00 = d11 / det, i01 = -d01 / det, i10 = -d10 / det, i11 = d00 / det;
      double T[2][2];
      for (int r = 0; r < 2; ++r) { T[r][0] = el(fl, r, 0) * i00 + el(fl, r, 1) * i10; T[r][1] = el(fl, r, 0) * i01 + el(fl, r, 1) * i11; }
      for (int r = 0; r < 2; ++r)
        for (int q = 0; q < 2; ++q) d[r * HS + (size_t)(fd / 16) * 2 + q] -= T[r][0] * el(fu, 0, q) + T[r][1] * el(fu, 1, q);
    }
    for (size_t i = 0; i < A.size(); ++i) A[i] += d[i];
  }
  for (int t = 0; t < gw; ++t) if ((unsigned)W[2 * (F.n_fwd * gw + t)] != INV) return -5;      // the padding pass
  if (F.n_scale != 0 || F.n_scale_rhs != 0) return -6;                    // no scaling pass: U and the right-hand side stay unscaled
  if (F.back_off % 2) return -8;
  auto inv_apply = [&](unsigned fd, double s0, double s1, double& x0, double& x1) {       // x = inv(D) s, D = block at field fd
    const double d00 = el(fd, 0, 0), d01 = el(fd, 0, 1), d10 = el(fd, 1, 0), d11 = el(fd, 1, 1);
    const double det = d00 * d11 - d01 * d10;
    x0 = (d11 * s0 - d01 * s1) / det;
    x1 = (d00 * s1 - d10 * s0) / det;
  };
  for (int k = 0; k < F.n_back; ++k) {
    std::vector<double> d(A.size(), 0.0);
    std::map<unsigned, int> dst_half;
    for (int t = 0; t < gw; ++t) {
      const unsigned w0 = (unsigned)W[F.back_off + 2 * (k * gw + t)], w1 = (unsigned)W[F.back_off + 2 * (k * gw + t) + 1];
      if (w0 == INV) continue;
      const unsigned fu = w0 & 0xffffu, fj = w0 >> 16;
      if (!ok_field(fu) || !ok_field(fj) || !ok_field(w1) || (int)fj < F.rhs_field0 || (int)w1 < F.rhs_field0) return -2;
      if (gw > 64) {
        auto it = dst_half.find(w1);
        if (it != dst_half.end() && it->second != t / 64) return -9;
        dst_half[w1] = t / 64;
      }
      double x0, x1;
      inv_apply(fj - (unsigned)F.rhs_field0, el(fj, 0, 0), el(fj, 1, 0), x0, x1);       // the item applies inv(D_col) itself
      for (int r = 0; r < 2; ++r) d[r * HS + (size_t)(w1 / 16) * 2] -= el(fu, r, 0) * x0 + el(fu, r, 1) * x1;
    }
    for (size_t i = 0; i < A.size(); ++i) A[i] += d[i];
  }
  for (int p = 0; p < n_sub; ++p) {                                       // the consumer forms x_p = inv(D_p) s_p
    double x0, x1;
    inv_apply((unsigned)p * 16u, A[((size_t)S.rslot0 + p) * 2], A[HS + ((size_t)S.rslot0 + p) * 2], x0, x1);
    x_out[p * 2] = x0; x_out[p * 2 + 1] = x1;
  }
  if (stats) { stats[0] = F.n_fwd; stats[1] = S.n_levels - S.gj_lv0; stats[2] = F.n_back; stats[3] = (int)F.words.size(); }
  return F.n_fwd + F.n_back;
}

// undirected off-diagonal pairs (build_upairs): out[4k + {0,1,2,3}] = u, v, row/col check flags; returns n_up or a negative error
extern "C" int sym_upairs_check(int n_sub, int n_line, const int* line_or, const int* line_ex) {
  gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, line_or, line_ex);
  const std::vector<int> up = gpf::build_upairs(S);
  const int n_up = (int)up.size() / 2;
  if (2 * n_up != S.nslot_y - S.n) return -1;                    // every off-diagonal block of the pattern belongs to one pair
  std::vector<char> seen(S.nslot_y, 0);
  for (int k = 0; k < n_up; ++k) {
    const int u = up[2 * k] & 0xffff, v = (int)((unsigned)up[2 * k] >> 16), suv = up[2 * k + 1] & 0xffff, svu = (int)((unsigned)up[2 * k + 1] >> 16);
    if (!(u < v) || suv < S.n || svu < S.n || suv >= S.nslot_y || svu >= S.nslot_y) return -2;
    if (S.slot_row[suv] != u || S.slot_col[suv] != v || S.slot_row[svu] != v || S.slot_col[svu] != u) return -3;
    if (seen[suv] || seen[svu]) return -4;
    seen[suv] = seen[svu] = 1;
  }
  return n_up;
}


// structural check of the level schedule (build_symbolic: greedy levels or the re-scheduled ones): returns 0 when (1) the pivots of a
// level are pairwise non-adjacent in the filled pattern, (2) every pivot that updates another one (a block (q, p) with p eliminated
// after q) sits in an EARLIER level, (3) every substation is a pivot exactly once; a negative code otherwise.  out[0] = levels,
// out[1] = height of the elimination DAG (the minimum any schedule with this fill can have), out[2] = passes of the flat program.
extern "C" int sym_check_schedule(int n_sub, int n_line, const int* line_or, const int* line_ex, int resched, int gw, int* out) {
  gpf::Symbolic S = gpf::build_symbolic(n_sub, n_line, line_or, line_ex, 1, -1, resched);
  std::vector<int> seen(n_sub, 0);
  for (int lv = 0; lv < S.n_levels; ++lv) {
    const int* h = S.prog.data() + (size_t)8 * lv;
    for (int k = 0; k < h[1]; ++k) { const int p = S.prog[h[0] + k]; if (p < 0 || p >= n_sub || seen[p]++) return -1; if (S.level_of[p] != lv) return -2; }
  }
  for (int p = 0; p < n_sub; ++p) if (seen[p] != 1) return -3;
  std::vector<int> depth(n_sub, 0);
  int height = 0;
  for (int lv = 0; lv < S.n_levels; ++lv)
    for (int p = 0; p < n_sub; ++p) {
      if (S.level_of[p] != lv) continue;
      for (int q = 0; q < S.nslot_lu; ++q) {
        if (S.slot_row[q] != p || S.slot_col[q] == p) continue;
        const int o = S.slot_col[q];
        if (S.level_of[o] == lv) return -4;                        // adjacent pivots in one level
        if (S.level_of[o] < lv) depth[p] = std::max(depth[p], depth[o] + 1);
      }
      height = std::max(height, depth[p] + 1);
    }
  if (S.n_levels < height) return -5;
  const gpf::FlatProg F = gpf::build_flat(S, gw, 0);
  if (out) { out[0] = S.n_levels; out[1] = height; out[2] = F.n_fwd + F.n_back; }
  return 0;
}
