// gridpf_small.hpp -- kernel v3 for small / medium grids (Newton unknowns n <= NMAX <= 64, active buses <= 64).
//
// Same pipeline and arithmetic as gridpf_kernels.hpp (one wavefront per grid instance) but built around what the
// per-phase cycle stamps (tools/phase_timing.py) showed to matter on MI355X for 14..36-bus grids: the kernel is bound
// by instruction issue and by dependent-latency chains, not by HBM.
//   * REGISTER-RESIDENT Gauss-Jordan with partial pivoting.  LPR SIMD lanes share one row of the compact Jacobian
//     [J | -F] (column j lives in sub-lane j % LPR), so a 22-unknown system occupies 44 lanes with 13 values each.
//     Pivot search = 6-step DPP max-reduction on a 32-bit key (high word of |a_ik|, lane id in the low 6 bits); the
//     pivot row is broadcast through a small LDS buffer (ds_write_b128 / ds_read_b128, in-order within the wave, no
//     barrier); candidate reciprocals are computed before the reduction so the divide is off the critical path; no
//     back-substitution (Gauss-Jordan).
//   * Jacobian rows are assembled by their LPR lanes (buses split between the sub-lanes, partial S combined with a
//     DPP quad permute) through a conflict-free LDS row buffer laid out exactly as the register file wants it.
//   * Ybus and B' are assembled by BRANCH lanes with native LDS f64 atomics (ds_add_f64) instead of per-bus gather
//     loops over global tables (41.8k -> ~2k cycles).
//   * The lane's injection row and topology row are staged in LDS once; per-bus data used by the assembly loop is
//     one packed 32-byte record (e, f, 1/|V|, pidx, qidx).
//   * Generator Q-split / slack-P reductions use v_readlane instead of loops over global tables.
//   * Branch-free Cody-Waite + fdlibm-kernel sincos; kernel parameters by pointer to a device-resident block.
#pragma once
#include "gridpf_kernels.hpp"

namespace gpf {

#ifdef GPF_TIMING
#define GPF_STAMP(k) do { if (tid == 0) P->b.work[(size_t)inst * 32 + (k)] = (double)(long long)__builtin_readcyclecounter(); } while (0)
#else
#define GPF_STAMP(k) do {} while (0)
#endif

#ifndef GPF_GJ_LDS_BCAST
#define GPF_GJ_LDS_BCAST 1
#endif

#ifndef GPF_MINW
#define GPF_MINW(NMAX) ((NMAX) <= 24 ? 4 : (NMAX) <= 32 ? 3 : 2)
#endif

struct DevParams {
  GridDev g;
  Bufs b;
  OutOff oo;
};

typedef signed char i8;

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// max over the 64 lanes of a 32-bit key (identity 0); the result is wave-uniform.
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// value held by sub-lane SRC of every group of LPR consecutive lanes (LPR = 1, 2 or 4: a DPP quad permute)
template <int LPR, int SRC>
__device__ __forceinline__ double group_bcast(double v) {
  if (LPR == 1) return v;
  constexpr int S2 = SRC & 1;
  constexpr int ctrl = (LPR == 2) ? (S2 | (S2 << 2) | ((2 + S2) << 4) | ((2 + S2) << 6))
                                  : ((SRC & 3) | ((SRC & 3) << 2) | ((SRC & 3) << 4) | ((SRC & 3) << 6));
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the LPR lanes of a group (every lane gets the total)
template <int LPR>
__device__ __forceinline__ double group_sum(double v) {
  if (LPR == 1) return v;
  {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, true);
    v += __hiloint2double(hi, lo);
  }
  if (LPR == 4) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xf, 0xf, true);
    v += __hiloint2double(hi, lo);
  }
  return v;
}

// 1/x to full double precision for normal, non-zero x (v_rcp_f64 + two Newton steps); no denormal / special-case
// handling -- callers test the pivot magnitude separately.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}

// sin / cos for |x| up to a few thousand radians: Cody-Waite reduction by pi/2 (3-part constant with FMA) +
// the fdlibm __kernel_sin / __kernel_cos minimax polynomials on [-pi/4, pi/4] (< 1 ulp).
__device__ __forceinline__ void fast_sincos(double x, double& s, double& c) {
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double P1 = 1.57079632673412561417e+00;
  const double P2 = 6.07710050650619224932e-11;
  const double P3 = 2.02226624879595063154e-21;
  const double kf = rint(x * TWO_OVER_PI);
  double r = fma(-kf, P1, x);
  r = fma(-kf, P2, r);
  r = fma(-kf, P3, r);
  const int q = (int)kf;
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
  const double sr = fma(r * z, ps, r);
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const bool swap = q & 1;
  double ss = swap ? cr : sr;
  double cc = swap ? sr : cr;
  if (q & 2) ss = -ss;
  if ((q + 1) & 2) cc = -cc;
  s = ss;
  c = cc;
}

// ---------------------------------------------------------------------------------------------------
// Geometry of the register-resident solver.
template <int NMAX, int LPR>
struct Geo {
  static_assert(NMAX % LPR == 0, "NMAX must be a multiple of LPR");
  static constexpr int NC = NMAX / LPR;                 // matrix columns per lane
  static constexpr int NV = (NC + 2) & ~1;              // values per lane (columns + rhs), padded to even (16-byte slots)
  static constexpr int RHS = NC;                        // slot of the right-hand side (every sub-lane keeps a copy)
  // position of column `col` of row `row` in the lane-major row buffer R
  __device__ static __forceinline__ int pos(int row, int col) { return (row * LPR + (col % LPR)) * NV + col / LPR; }
};

// Register-resident Gauss-Jordan with partial pivoting.  lane = row*LPR + sub; a[idx] = column idx*LPR+sub of the
// row (idx < NC), a[RHS] = right-hand side.  Rows >= n are idle.  `pb` = LDS pivot-row buffer (LPR*NV doubles).
// On return every lane of the row that pivoted column k has mycol = k and x = x_k.  Returns false on a zero pivot.
template <int NMAX, int LPR, int NV>
__device__ __forceinline__ bool gj_solve(double (&a)[NV], int n, int lane, double* __restrict__ pb, double& x, int& mycol) {
  using G = Geo<NMAX, LPR>;
  static_assert(NV == G::NV, "register row size mismatch");
  const int row = lane / LPR, sub = lane % LPR;
  bool used = row >= n;
  bool ok = true;
  double inv_piv = 1.0;
  mycol = -1;
  double* my_pb = pb + sub * NV;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k < n) {   // wave-uniform
      const int ks = k % LPR, ki = k / LPR;
      const bool owner = (LPR == 1) || (sub == ks);
      const double akk = a[ki];
      unsigned key = (((unsigned)__double2hiint(fabs(akk))) & ~63u) | (unsigned)lane;
      if (used || !owner) key = (unsigned)lane;
      const double cand = fast_rcp(akk);                 // independent of the reduction: overlaps with it
      const unsigned best = wave_umax(key);
      const int pl = (int)(best & 63u);
      if ((best >> 6) == 0u) ok = false;
      const double rp = readlane_f64(cand, pl);
      const double m_own = akk * rp;
      double m = (LPR == 1) ? m_own
                            : (ks == 0 ? group_bcast<LPR, 0>(m_own)
                                       : ks == 1 ? group_bcast<LPR, 1>(m_own)
                                                 : ks == 2 ? group_bcast<LPR, 2>(m_own) : group_bcast<LPR, 3>(m_own));
      const bool in_prow = (row == pl / LPR);
      if (in_prow) { used = true; mycol = k; inv_piv = rp; m = 0.0; }
      if (LPR == 1) {
        // one lane per row: the pivot lane is wave-uniform -> v_readlane straight into SGPR operands of the FMAs
#pragma unroll
        for (int idx = ki; idx < NV - 1; ++idx) a[idx] = fma(-m, readlane_f64(a[idx], pl), a[idx]);
      } else {
#if GPF_GJ_LDS_BCAST
      if (in_prow) {
        // publish the live part of the pivot row (slots >= ki) for the other rows
#pragma unroll
        for (int idx = ki & ~1; idx < NV; idx += 2) {
          double2 v2; v2.x = a[idx]; v2.y = a[idx + 1];
          *reinterpret_cast<double2*>(my_pb + idx) = v2;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int idx = ki & ~1; idx < NV; idx += 2) {
        const double2 v2 = *reinterpret_cast<const double2*>(my_pb + idx);
        a[idx] = fma(-m, v2.x, a[idx]);
        a[idx + 1] = fma(-m, v2.y, a[idx + 1]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#else
      // pull the live part of the pivot row (slots >= ki) from the lane of the pivot row that holds the same
      // column class: ds_bpermute (LDS crossbar, no LDS memory traffic, no write->read dependency)
      const int src = ((pl / LPR) * LPR + sub) << 2;
#pragma unroll
      for (int idx = ki; idx < NV - 1; ++idx) {
        const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(a[idx]));
        const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(a[idx]));
        a[idx] = fma(-m, __hiloint2double(hi, lo), a[idx]);
      }
#endif
      }
    }
  }
  x = a[G::RHS] * inv_piv;
  return ok;
}

// ---------------------------------------------------------------------------------------------------
struct BusRec {      // per-bus record read by the Jacobian assembly loop (two ds_read_b128)
  double e, f, ivm;
  int pidx, qidx;
};

struct CarveS {
  double* R;         // [<=64 lanes][NV] lane-major row buffer (Jacobian / B' rows); its head is reused as pivot-row
                     // buffer, solution vector and (after the Newton loop) bus injections
  double* Y;         // [nbc][ldy] complex (re, im interleaved), ldy = nbc | 1
  BusRec* rec;       // [nbc]
  double *vm, *va, *Psp, *Qsp, *vset, *Pd, *Qd, *Gs;     // [nbc]
  double* inj;       // [n_inj] staged injection row of the lane
  double *dx, *Sre, *Sim, *pb;                           // aliases inside R
  int *btype, *lab;  // [nbc]
  int* topo;         // [dim_topo] staged topology row (aliases R during K1)
  i8* gmap;          // [nb_tot]
  i8 *pbus, *qbus;   // [NMAX]
  i8 *lor_c, *lex_c, *gen_c, *load_c, *sto_c, *sh_c;
};

template <int NMAX, int LPR>
__host__ __device__ inline size_t rbuf_doubles(const GridDev& g, int nbc, int nrows) {
  using G = Geo<NMAX, LPR>;
  size_t rbuf = (size_t)nrows * LPR * G::NV;
  const size_t min_r = (size_t)LPR * G::NV + NMAX + 2 * (size_t)nbc;       // pivot buffer + dx + Sre/Sim
  if (rbuf < min_r) rbuf = min_r;
  const size_t topo_d = ((size_t)g.dim_topo + 1) / 2 + 1;
  if (rbuf < topo_d) rbuf = topo_d;
  return (rbuf + 1) & ~(size_t)1;
}

template <int NMAX, int LPR>
__host__ __device__ inline size_t lds_bytes_small(const GridDev& g, int nbc, int nrows) {
  const size_t ldy = (size_t)(nbc | 1);
  const size_t nd = rbuf_doubles<NMAX, LPR>(g, nbc, nrows) + 2 * (size_t)nbc * ldy + 4 * (size_t)nbc /*BusRec*/ + 8 * (size_t)nbc +
                    (size_t)g.n_inj;
  const size_t ni = 2 * (size_t)nbc;
  const size_t n8 = (size_t)g.nb_tot + 2 * (size_t)NMAX + 2 * (size_t)g.n_line + g.n_gen + g.n_load + g.n_sto + g.n_shunt;
  return nd * 8 + ni * 4 + ((n8 + 15) & ~(size_t)15);
}

template <int NMAX, int LPR>
__device__ inline void carve_small(CarveS& c, unsigned char* base, const GridDev& g, int nbc, int nrows) {
  using G = Geo<NMAX, LPR>;
  const int ldy = nbc | 1;
  double* d = reinterpret_cast<double*>(base);
  c.R = d;
  c.pb = d;                                  // pivot-row buffer: first LPR*NV doubles
  c.dx = d + LPR * G::NV;                    // solution vector: next NMAX doubles
  c.Sre = d + LPR * G::NV + NMAX;            // bus injections (only live after the Newton loop)
  c.Sim = c.Sre + nbc;
  c.topo = reinterpret_cast<int*>(d);        // staged topology row (only live during K1)
  d += rbuf_doubles<NMAX, LPR>(g, nbc, nrows);
  c.Y = d; d += (size_t)2 * nbc * ldy;
  c.rec = reinterpret_cast<BusRec*>(d); d += (size_t)4 * nbc;
  c.vm = d; d += nbc; c.va = d; d += nbc; c.Psp = d; d += nbc; c.Qsp = d; d += nbc;
  c.vset = d; d += nbc; c.Pd = d; d += nbc; c.Qd = d; d += nbc; c.Gs = d; d += nbc;
  c.inj = d; d += g.n_inj;
  int* i = reinterpret_cast<int*>(d);
  c.btype = i; i += nbc; c.lab = i; i += nbc;
  i8* q = reinterpret_cast<i8*>(i);
  c.gmap = q; q += g.nb_tot;
  c.pbus = q; q += NMAX; c.qbus = q; q += NMAX;
  c.lor_c = q; q += g.n_line; c.lex_c = q; q += g.n_line;
  c.gen_c = q; q += g.n_gen; c.load_c = q; q += g.n_load; c.sto_c = q; q += g.n_sto; c.sh_c = q; q += g.n_shunt;
}

// ---------------------------------------------------------------------------------------------------
// K1 (small): element -> compact bus maps (int8), active-bus mask, dense renumbering.  Needs nb_tot <= 127.
__device__ inline int build_topology_small(const GridDev& g, CarveS& c, const int* __restrict__ topo_g,
                                           const int* __restrict__ shunt_bus, unsigned char* __restrict__ status_out, int nbc,
                                           int tid) {
  const int ns = g.n_sub;
  for (int i = tid; i < g.dim_topo; i += WAVE) c.topo[i] = topo_g[i];     // coalesced row load, then LDS gathers
  for (int i = tid; i < g.nb_tot; i += WAVE) c.gmap[i] = 0;
  __syncthreads();
  const int* topo = c.topo;
  for (int l = tid; l < g.n_line; l += WAVE) {
    const int bo = topo[g.line_or_pos[l]], be = topo[g.line_ex_pos[l]];
    const bool on = (bo >= 1) && (be >= 1);
    const int go = on ? g.line_or_sub[l] + (bo - 1) * ns : -1;
    const int ge = on ? g.line_ex_sub[l] + (be - 1) * ns : -1;
    c.lor_c[l] = (i8)go;
    c.lex_c[l] = (i8)ge;
    if (on) { c.gmap[go] = 1; c.gmap[ge] = 1; }
    if (status_out) status_out[l] = on ? 1 : 0;
  }
  for (int i = tid; i < g.n_gen; i += WAVE) {
    const int b = topo[g.gen_pos[i]];
    const int gb = (b >= 1) ? g.gen_sub[i] + (b - 1) * ns : -1;
    c.gen_c[i] = (i8)gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  for (int i = tid; i < g.n_load; i += WAVE) {
    const int b = topo[g.load_pos[i]];
    const int gb = (b >= 1) ? g.load_sub[i] + (b - 1) * ns : -1;
    c.load_c[i] = (i8)gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  for (int i = tid; i < g.n_sto; i += WAVE) {
    const int b = topo[g.sto_pos[i]];
    const int gb = (b >= 1) ? g.sto_sub[i] + (b - 1) * ns : -1;
    c.sto_c[i] = (i8)gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  for (int i = tid; i < g.n_shunt; i += WAVE) {
    const int b = shunt_bus[i];
    const int gb = (b >= 1) ? g.shunt_sub[i] + (b - 1) * ns : -1;
    c.sh_c[i] = (i8)gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  __syncthreads();
  int base = 0;
  for (int i0 = 0; i0 < g.nb_tot; i0 += WAVE) {
    const int i = i0 + tid;
    const int act = (i < g.nb_tot) ? (int)c.gmap[i] : 0;
    const unsigned long long m = __ballot(act);
    const int rank = base + __popcll(m & ((1ull << tid) - 1ull));
    if (i < g.nb_tot) c.gmap[i] = (i8)(act ? rank : -1);
    base += __popcll(m);
  }
  __syncthreads();
  const int nb = base;
  if (nb > nbc) return -1;
  for (int l = tid; l < g.n_line; l += WAVE) {
    const int go = c.lor_c[l], ge = c.lex_c[l];
    c.lor_c[l] = go >= 0 ? c.gmap[go] : (i8)-1;
    c.lex_c[l] = ge >= 0 ? c.gmap[ge] : (i8)-1;
  }
  for (int i = tid; i < g.n_gen; i += WAVE) { const int b = c.gen_c[i]; c.gen_c[i] = b >= 0 ? c.gmap[b] : (i8)-1; }
  for (int i = tid; i < g.n_load; i += WAVE) { const int b = c.load_c[i]; c.load_c[i] = b >= 0 ? c.gmap[b] : (i8)-1; }
  for (int i = tid; i < g.n_sto; i += WAVE) { const int b = c.sto_c[i]; c.sto_c[i] = b >= 0 ? c.gmap[b] : (i8)-1; }
  for (int i = tid; i < g.n_shunt; i += WAVE) { const int b = c.sh_c[i]; c.sh_c[i] = b >= 0 ? c.gmap[b] : (i8)-1; }
  __syncthreads();
  return nb;
}

// ---------------------------------------------------------------------------------------------------
// One complete power flow of one instance.  `inj_staged`: c.inj already holds the lane's injection row.
template <int NMAX, int LPR>
__device__ inline int solve_instance_small(const DevParams* __restrict__ P, CarveS& c, int inst, int nbc, int nrows, int is_dc,
                                           int max_iter, double tol_pu, int tid, bool inj_staged, int& n_iter_out, int& nb_out) {
  using G = Geo<NMAX, LPR>;
  constexpr int NV = G::NV;
  const GridDev& g = P->g;
  const Bufs& b = P->b;
  const OutOff& oo = P->oo;
  const int* __restrict__ topo_g = b.topo + (size_t)inst * g.dim_topo;
  const int* __restrict__ shb = b.shunt_bus + (size_t)inst * g.n_shunt;
  unsigned char* lstat = b.line_status + (size_t)inst * g.n_line;
  n_iter_out = 0;
  nb_out = 0;
  GPF_STAMP(0);
  if (!inj_staged) {
    const double* __restrict__ inj_g = b.inj + (size_t)inst * g.n_inj;
    for (int i = tid; i < g.n_inj; i += WAVE) c.inj[i] = inj_g[i];
  }
  const double* __restrict__ inj = c.inj;

  // ---- K1 ----------------------------------------------------------------------------------------------------------
  const int nb = build_topology_small(g, c, topo_g, shb, lstat, nbc, tid);
  if (nb < 0) return 5;
  nb_out = nb;
  const int ldy = nbc | 1;
  const double sn = g.sn_mva;
  const double inv_sn = 1.0 / sn;
  GPF_STAMP(1);

  // ---- bus types / injections: thread per bus gathers over the (LDS-staged) elements, oracle summation order --------
  for (int ci = tid; ci < nb; ci += WAVE) {
    int bt = BT_PQ;
    double Pg = 0.0, vs = 1.0;
    for (int i = 0; i < g.n_gen; ++i) {
      const bool hit = c.gen_c[i] == ci;
      const bool sl = g.gen_slack[i] != 0;          // wave-uniform
      const double gp = inj[oo.inj_gen_p + i], gv = inj[oo.inj_gen_vm + i];
      if (hit) {
        if (sl) bt = BT_REF;
        else { if (bt != BT_REF) bt = BT_PV; Pg += gp * inv_sn; }
        vs = gv;
      }
    }
    double pd = 0.0, qd = 0.0;
    for (int i = 0; i < g.n_load; ++i) {
      const bool hit = c.load_c[i] == ci;
      const double lp = inj[oo.inj_load_p + i], lq = inj[oo.inj_load_q + i];
      if (hit) { pd += lp; qd += lq; }
    }
    for (int i = 0; i < g.n_sto; ++i) {
      const bool hit = c.sto_c[i] == ci;
      const double sp = inj[oo.inj_sto_p + i], sq = inj[oo.inj_sto_q + i];
      if (hit) { pd += sp; qd += sq; }
    }
    double gs = 0.0;
    for (int i = 0; i < g.n_shunt; ++i) {
      const bool hit = c.sh_c[i] == ci;
      const double sp = inj[oo.inj_sh_p + i] * g.shunt_fact[i] * inv_sn;
      if (hit) gs += sp;
    }
    c.btype[ci] = bt;
    c.vset[ci] = vs;
    c.Pd[ci] = pd;
    c.Qd[ci] = qd;
    c.Gs[ci] = gs;
    c.Psp[ci] = Pg - pd * inv_sn;
    c.Qsp[ci] = -qd * inv_sn;
    c.lab[ci] = (bt == BT_REF) ? 1 : 0;
  }
  __syncthreads();
  int npvpq = 0, npq = 0, nref = 0;
  for (int i0 = 0; i0 < nb; i0 += WAVE) {
    const int ci = i0 + tid;
    const int bt = (ci < nb) ? c.btype[ci] : -1;
    const unsigned long long mp = __ballot(bt == BT_PQ || bt == BT_PV);
    const unsigned long long mq = __ballot(bt == BT_PQ);
    const unsigned long long mr = __ballot(bt == BT_REF);
    const unsigned long long below = (1ull << tid) - 1ull;
    if (ci < nb) {
      const int pi = (bt == BT_PQ || bt == BT_PV) ? npvpq + __popcll(mp & below) : -1;
      const int qi = (bt == BT_PQ) ? npq + __popcll(mq & below) : -1;
      c.rec[ci].pidx = pi;
      c.rec[ci].qidx = qi;
      if (pi >= 0 && pi < NMAX) c.pbus[pi] = (i8)ci;
      if (qi >= 0 && qi < NMAX) c.qbus[qi] = (i8)ci;
    }
    npvpq += __popcll(mp);
    npq += __popcll(mq);
    nref += __popcll(mr);
  }
  __syncthreads();
  if (nref == 0) return 3;
  const int n = npvpq + npq;
  if (n > NMAX || n > nrows || n * LPR > WAVE) return 5;
  GPF_STAMP(2);

  // ---- connectivity ----------------------------------------------------------------------------------------------------
  for (int sweep = 0; sweep < nb; ++sweep) {
    int changed = 0;
    for (int l = tid; l < g.n_line; l += WAVE) {
      const int f = c.lor_c[l], t = c.lex_c[l];
      if (f >= 0) {
        const int lf = c.lab[f], lt = c.lab[t];
        if (lf != lt) { c.lab[f] = 1; c.lab[t] = 1; changed = 1; }
      }
    }
    __syncthreads();
    if (!__any(changed)) break;
  }
  {
    int bad = 0;
    for (int ci = tid; ci < nb; ci += WAVE) bad |= (c.lab[ci] == 0);
    if (__any(bad)) return 2;
  }
  GPF_STAMP(3);

  // ---- K2 + K3 assembly with LDS f64 atomics from the BRANCH lanes -----------------------------------------------------------
  for (int e = tid; e < 2 * nb * ldy; e += WAVE) c.Y[e] = 0.0;
  if (tid < npvpq * LPR) {
    double* Rz = c.R + (size_t)tid * NV;
#pragma unroll
    for (int idx = 0; idx < NV; ++idx) Rz[idx] = 0.0;
  }
  __syncthreads();
  for (int l = tid; l < g.n_line; l += WAVE) {
    const int f = c.lor_c[l], t = c.lex_c[l];
    if (f < 0) continue;
    if (!is_dc) {
      const double4* y4 = reinterpret_cast<const double4*>(g.br_y + (size_t)8 * l);
      const double4 ya = y4[0], yb = y4[1];          // yff.re, yff.im, yft.re, yft.im | ytf.re, ytf.im, ytt.re, ytt.im
      double* Yf = c.Y + (size_t)2 * f * ldy;
      double* Yt = c.Y + (size_t)2 * t * ldy;
      atomicAdd(&Yf[2 * f], ya.x); atomicAdd(&Yf[2 * f + 1], ya.y);
      atomicAdd(&Yf[2 * t], ya.z); atomicAdd(&Yf[2 * t + 1], ya.w);
      atomicAdd(&Yt[2 * f], yb.x); atomicAdd(&Yt[2 * f + 1], yb.y);
      atomicAdd(&Yt[2 * t], yb.z); atomicAdd(&Yt[2 * t + 1], yb.w);
    }
    if (f != t) {
      const double bb = g.br_bdc[l];
      const int pf = c.rec[f].pidx, pt = c.rec[t].pidx;
      if (pf >= 0) atomicAdd(&c.R[G::pos(pf, pf)], bb);
      if (pt >= 0) atomicAdd(&c.R[G::pos(pt, pt)], bb);
      if (pf >= 0 && pt >= 0) { atomicAdd(&c.R[G::pos(pf, pt)], -bb); atomicAdd(&c.R[G::pos(pt, pf)], -bb); }
    }
  }
  if (!is_dc) {
    for (int s = tid; s < g.n_shunt; s += WAVE) {
      const int ci = c.sh_c[s];
      if (ci >= 0) {
        const double fct = g.shunt_fact[s] * inv_sn;
        atomicAdd(&c.Y[(size_t)2 * ci * ldy + 2 * ci], inj[oo.inj_sh_p + s] * fct);
        atomicAdd(&c.Y[(size_t)2 * ci * ldy + 2 * ci + 1], -inj[oo.inj_sh_q + s] * fct);
      }
    }
  }
  for (int ci = tid; ci < nb; ci += WAVE) {           // DC right-hand side (a copy in every sub-lane segment)
    const int pi = c.rec[ci].pidx;
    if (pi >= 0) {
      const double rhs = c.Psp[ci] - c.Gs[ci];
#pragma unroll
      for (int s = 0; s < LPR; ++s) c.R[(size_t)(pi * LPR + s) * NV + G::RHS] = rhs;
    }
  }
  __syncthreads();
  GPF_STAMP(4);

  double a[NV];
  // ---- K3: DC solve ------------------------------------------------------------------------------------------------------
  {
    const bool on = tid < npvpq * LPR;
    const double* Rl = c.R + (size_t)(on ? tid : 0) * NV;
#pragma unroll
    for (int idx = 0; idx < NV; idx += 2) {
      const double2 v2 = *reinterpret_cast<const double2*>(Rl + idx);
      a[idx] = on ? v2.x : 0.0;
      a[idx + 1] = on ? v2.y : 0.0;
    }
    __syncthreads();                 // everybody holds its row: the buffer head may now be reused (pivot rows, dx)
    double x;
    int mycol;
    bool ok = gj_solve<NMAX, LPR, NV>(a, npvpq, tid, c.pb, x, mycol);
    if (mycol >= 0 && (tid % LPR) == 0) {
      c.dx[mycol] = x;
      if (!(fabs(x) < 1e300)) ok = false;
    }
    __syncthreads();
    if (__any(!ok)) return 4;
    for (int ci = tid; ci < nb; ci += WAVE) {
      const int pi = c.rec[ci].pidx;
      c.va[ci] = (pi >= 0) ? c.dx[pi] : 0.0;
      c.vm[ci] = (c.btype[ci] == BT_PQ) ? 1.0 : c.vset[ci];
    }
    __syncthreads();
  }
  GPF_STAMP(5);

  int status = 0;
  int it = 0;
  if (!is_dc) {
    // ---- K4/K5: Newton-Raphson ---------------------------------------------------------------------------------------------
    bool converged = false;
    const int row = tid / LPR, sub = tid % LPR;
    const bool row_on = row < n;
    const bool isQ = row >= npvpq;
    const int ib = row_on ? (isQ ? (int)c.qbus[row - npvpq] : (int)c.pbus[row]) : 0;
    const double Pi = c.Psp[ib], Qi = c.Qsp[ib];
    const int pii = c.rec[ib].pidx, qii = c.rec[ib].qidx;
    double* Rrow = c.R + (size_t)(row_on ? row : 0) * LPR * NV;    // this row's LPR segments
    const double* Rl = c.R + (size_t)(row_on ? tid : 0) * NV;      // this lane's segment
    double Sr_last = 0.0, Si_last = 0.0;
    while (true) {
      for (int ci = tid; ci < nb; ci += WAVE) {
        double s, co;
        fast_sincos(c.va[ci], s, co);
        const double vmi = c.vm[ci];
        c.rec[ci].e = vmi * co;
        c.rec[ci].f = vmi * s;
        c.rec[ci].ivm = 1.0 / vmi;
      }
      __syncthreads();
      if (it == 1) GPF_STAMP(10);
      // Row assembly: T_ij = V_i conj(Y_ij V_j), S_i = sum_j T_ij; the LPR lanes of a row split the buses j
      double fabs_mis = 0.0;
      bool bad = false;
      double Sr = 0.0, Si = 0.0, Tr = 0.0, Ti = 0.0;
      if (row_on) {
        const double ei = c.rec[ib].e, fi = c.rec[ib].f;
        const double* Yr = c.Y + (size_t)2 * ib * ldy;
#pragma unroll 2
        for (int j = sub; j < nb; j += LPR) {
          const double2 y = *reinterpret_cast<const double2*>(Yr + 2 * j);
          const BusRec rj = c.rec[j];
          const double aa = y.x * rj.e - y.y * rj.f, bb = y.x * rj.f + y.y * rj.e;
          const double tr_ = ei * aa + fi * bb;
          const double ti_ = fi * aa - ei * bb;
          Sr += tr_;
          Si += ti_;
          if (rj.pidx >= 0) Rrow[(rj.pidx % LPR) * NV + rj.pidx / LPR] = isQ ? -tr_ : ti_;
          if (rj.qidx >= 0) {
            const int cq = npvpq + rj.qidx;
            Rrow[(cq % LPR) * NV + cq / LPR] = (isQ ? ti_ : tr_) * rj.ivm;
          }
          if (j == ib) { Tr = tr_; Ti = ti_; }
        }
      }
      Sr = group_sum<LPR>(Sr);
      Si = group_sum<LPR>(Si);
      if (row_on) {
        if ((ib % LPR) == sub) {
          // diagonal block: dS/dVa_ii = j (S - T_ii), dS/dVm_ii = (T_ii + S) / |V_i|
          const double ivmi = c.rec[ib].ivm;
          Rrow[(pii % LPR) * NV + pii / LPR] = isQ ? (Sr - Tr) : (Ti - Si);
          if (qii >= 0) {
            const int cq = npvpq + qii;
            Rrow[(cq % LPR) * NV + cq / LPR] = (isQ ? (Ti + Si) : (Tr + Sr)) * ivmi;
          }
        }
        const double mis = isQ ? (Si - Qi) : (Sr - Pi);
        Rrow[sub * NV + G::RHS] = -mis;                 // every sub-lane keeps a copy of the right-hand side
        fabs_mis = fabs(mis);
        if (!(fabs_mis <= 1e300)) bad = true;
        Sr_last = Sr;
        Si_last = Si;
      }
      // ||F||inf < tol  <=>  no row has |F_row| >= tol
      const bool any_ge = __any(row_on && !(fabs_mis < tol_pu));
      if (__any(bad)) { status = 1; break; }
      if (!any_ge) { converged = true; break; }
      if (it >= max_iter) break;
      ++it;
      __syncthreads();               // the LPR lanes of a row wrote each other's segments
      if (it == 2) GPF_STAMP(11);
#pragma unroll
      for (int idx = 0; idx < NV; idx += 2) {
        const double2 v2 = *reinterpret_cast<const double2*>(Rl + idx);
        // columns >= n of the compact system are not part of it: zero them (slot RHS is always live)
        const int c0 = idx * LPR + sub, c1 = (idx + 1) * LPR + sub;
        a[idx] = (row_on && (c0 < n || idx == G::RHS)) ? v2.x : 0.0;
        a[idx + 1] = (row_on && (c1 < n || idx + 1 == G::RHS)) ? v2.y : 0.0;
      }
      __syncthreads();
      double x;
      int mycol;
      bool ok = gj_solve<NMAX, LPR, NV>(a, n, tid, c.pb, x, mycol);
      if (mycol >= 0 && sub == 0) {
        c.dx[mycol] = x;
        if (!(fabs(x) < 1e300)) ok = false;
      }
      __syncthreads();
      if (it == 2) GPF_STAMP(12);
      if (__any(!ok)) { status = 4; break; }
      for (int ci = tid; ci < nb; ci += WAVE) {
        const int pi = c.rec[ci].pidx, qi = c.rec[ci].qidx;
        double va = c.va[ci], vm = c.vm[ci];
        if (pi >= 0) va += c.dx[pi];
        if (qi >= 0) vm += c.dx[npvpq + qi];
        if (vm < 0.0) { vm = -vm; va += 3.14159265358979323846; }
        if (fabs(va) > 3.14159265358979323846) va = remainder(va, 6.28318530717958647692);
        c.va[ci] = va;
        c.vm[ci] = vm;
      }
      __syncthreads();
    }
    if (status == 0 && !converged) status = 1;
    __syncthreads();                       // the row buffer is dead from here on: part of it becomes Sre / Sim
    if (row_on && !isQ && sub == 0) { c.Sre[ib] = Sr_last; c.Sim[ib] = Si_last; }
  }
  n_iter_out = it;
  if (status != 0) return status;
  GPF_STAMP(6);

  // ---- K6: result extraction -----------------------------------------------------------------------------------------------
  float* out = b.out + (size_t)inst * g.n_out;
  const double RAD2DEG = 57.295779513082320877;
  const double SQRT3 = 1.7320508075688772935;
  __syncthreads();
  if (is_dc) {
    for (int ci = tid; ci < nb; ci += WAVE) {
      double acc = 0.0;
      for (int l = 0; l < g.n_line; ++l) {
        const int f = c.lor_c[l], t = c.lex_c[l];
        if (f < 0) continue;
        if (f == ci) acc += (c.va[f] - c.va[t]) * g.br_bdc[l];
        if (t == ci) acc -= (c.va[f] - c.va[t]) * g.br_bdc[l];
      }
      c.Sre[ci] = acc + c.Gs[ci];
      c.Sim[ci] = 0.0;
    }
  } else {
    // bus injections of the buses that own no Jacobian row (reference buses): S = V conj(Ybus V)
    for (int ci = tid; ci < nb; ci += WAVE) {
      if (c.rec[ci].pidx >= 0) continue;
      const double* Yr = c.Y + (size_t)2 * ci * ldy;
      double ir = 0.0, ii = 0.0;
      for (int j = 0; j < nb; ++j) {
        const double yr = Yr[2 * j], yi = Yr[2 * j + 1];
        ir += yr * c.rec[j].e - yi * c.rec[j].f;
        ii += yr * c.rec[j].f + yi * c.rec[j].e;
      }
      c.Sre[ci] = c.rec[ci].e * ir + c.rec[ci].f * ii;
      c.Sim[ci] = c.rec[ci].f * ir - c.rec[ci].e * ii;
    }
  }
  __syncthreads();
  for (int l = tid; l < g.n_line; l += WAVE) {
    const int f = c.lor_c[l], t = c.lex_c[l];
    float p_or = 0.f, q_or = 0.f, v_or = 0.f, a_or = 0.f, th_or = 0.f;
    float p_ex = 0.f, q_ex = 0.f, v_ex = 0.f, a_ex = 0.f, th_ex = 0.f;
    if (f >= 0) {
      const double vnf = g.sub_vn_kv[g.line_or_sub[l]], vnt = g.sub_vn_kv[g.line_ex_sub[l]];
      const double vmf = c.vm[f], vmt = c.vm[t];
      double pf, qf, pt, qt;
      if (is_dc) {
        pf = (c.va[f] - c.va[t]) * g.br_bdc[l] * sn;
        pt = -pf; qf = 0.0; qt = 0.0;
      } else {
        const double4* y4 = reinterpret_cast<const double4*>(g.br_y + (size_t)8 * l);
        const double4 ya = y4[0], yb = y4[1];
        const double ef = c.rec[f].e, ff = c.rec[f].f, et = c.rec[t].e, ft = c.rec[t].f;
        const double ifr = ya.x * ef - ya.y * ff + ya.z * et - ya.w * ft;
        const double ifi = ya.x * ff + ya.y * ef + ya.z * ft + ya.w * et;
        const double itr = yb.x * ef - yb.y * ff + yb.z * et - yb.w * ft;
        const double iti = yb.x * ff + yb.y * ef + yb.z * ft + yb.w * et;
        pf = (ef * ifr + ff * ifi) * sn;  qf = (ff * ifr - ef * ifi) * sn;
        pt = (et * itr + ft * iti) * sn;  qt = (ft * itr - et * iti) * sn;
      }
      p_or = (float)pf; q_or = (float)qf; p_ex = (float)pt; q_ex = (float)qt;
      a_or = (float)(sqrt(pf * pf + qf * qf) / (SQRT3 * vmf * vnf) * 1000.0);
      a_ex = (float)(sqrt(pt * pt + qt * qt) / (SQRT3 * vmt * vnt) * 1000.0);
      v_or = (float)(vmf * vnf); v_ex = (float)(vmt * vnt);
      th_or = (float)(c.va[f] * RAD2DEG); th_ex = (float)(c.va[t] * RAD2DEG);
    }
    out[oo.p_or + l] = p_or; out[oo.q_or + l] = q_or; out[oo.v_or + l] = v_or; out[oo.a_or + l] = a_or; out[oo.th_or + l] = th_or;
    out[oo.p_ex + l] = p_ex; out[oo.q_ex + l] = q_ex; out[oo.v_ex + l] = v_ex; out[oo.a_ex + l] = a_ex; out[oo.th_ex + l] = th_ex;
  }
  for (int i = tid; i < g.n_load; i += WAVE) {
    const int ci = c.load_c[i];
    const bool on = ci >= 0;
    out[oo.load_p + i] = on ? (float)inj[oo.inj_load_p + i] : 0.f;
    out[oo.load_q + i] = (on && !is_dc) ? (float)inj[oo.inj_load_q + i] : 0.f;
    out[oo.load_v + i] = on ? (float)(c.vm[ci] * g.sub_vn_kv[g.load_sub[i]]) : 0.f;
    out[oo.load_th + i] = on ? (float)(c.va[ci] * RAD2DEG) : 0.f;
  }
  for (int i = tid; i < g.n_sto; i += WAVE) {
    const int ci = c.sto_c[i];
    const bool on = ci >= 0;
    out[oo.sto_p + i] = on ? (float)inj[oo.inj_sto_p + i] : 0.f;
    out[oo.sto_q + i] = (on && !is_dc) ? (float)inj[oo.inj_sto_q + i] : 0.f;
    out[oo.sto_v + i] = on ? (float)(c.vm[ci] * g.sub_vn_kv[g.sto_sub[i]]) : 0.f;
    out[oo.sto_th + i] = on ? (float)(c.va[ci] * RAD2DEG) : 0.f;
  }
  int* sbo = b.shunt_bus_out + (size_t)inst * g.n_shunt;
  for (int i = tid; i < g.n_shunt; i += WAVE) {
    const int ci = c.sh_c[i];
    const bool on = ci >= 0;
    const double v = on ? c.vm[ci] : 0.0;
    out[oo.sh_p + i] = on ? (float)(inj[oo.inj_sh_p + i] * g.shunt_fact[i] * v * v) : 0.f;
    out[oo.sh_q + i] = (on && !is_dc) ? (float)(inj[oo.inj_sh_q + i] * g.shunt_fact[i] * v * v) : 0.f;
    out[oo.sh_v + i] = on ? (float)(v * g.sub_vn_kv[g.shunt_sub[i]]) : 0.f;
    sbo[i] = on ? shb[i] : -1;
  }
  // generators: pypower pfsoln.  Lane i owns generator i (+64, ...); the per-bus totals are accumulated with
  // v_readlane over the generators (uniform loop, no table reloads).
  for (int i0 = 0; i0 < g.n_gen; i0 += WAVE) {
    const int i = i0 + tid;
    const bool have = i < g.n_gen;
    const int ci = have ? (int)c.gen_c[i] : -1;
    const double my_minq = have ? g.gen_min_q[i] : 0.0, my_maxq = have ? g.gen_max_q[i] : 0.0;
    const int my_slack = have ? (int)g.gen_slack[i] : 0;
    const double my_p = have ? inj[oo.inj_gen_p + i] : 0.0;
    int cnt = 0, nslack = 0;
    double qmin_t = 0.0, qmax_t = 0.0, p_others = 0.0;
    for (int k0 = 0; k0 < g.n_gen; k0 += WAVE) {
      const int k_ = k0 + tid;
      const bool hk = k_ < g.n_gen;
      const int kc = hk ? (int)c.gen_c[k_] : -2;
      const double kminq = hk ? g.gen_min_q[k_] : 0.0, kmaxq = hk ? g.gen_max_q[k_] : 0.0;
      const int ksl = hk ? (int)g.gen_slack[k_] : 0;
      const double kp = hk ? inj[oo.inj_gen_p + k_] : 0.0;
      const int kn = min(WAVE, g.n_gen - k0);
      for (int kk = 0; kk < kn; ++kk) {
        const int bc = __builtin_amdgcn_readlane(kc, kk);
        const double bminq = readlane_f64(kminq, kk), bmaxq = readlane_f64(kmaxq, kk), bp = readlane_f64(kp, kk);
        const int bsl = __builtin_amdgcn_readlane(ksl, kk);
        if (bc == ci && ci >= 0) {
          ++cnt;
          qmin_t += bminq;
          qmax_t += bmaxq;
          if (bsl) ++nslack; else p_others += bp;
        }
      }
    }
    float gp = 0.f, gq = 0.f, gv = 0.f, gth = 0.f;
    if (ci >= 0) {
      const double qtot = c.Sim[ci] * sn + c.Qd[ci];
      double q;
      if (is_dc) q = 0.0;
      else if (cnt == 1) q = qtot;
      else if (qmin_t == qmax_t) q = qtot / cnt;
      else q = my_minq + (qtot - qmin_t) / (qmax_t - qmin_t + 2.220446049250313e-16) * (my_maxq - my_minq);
      double p = my_p;
      if (my_slack) p = (c.Sre[ci] * sn + c.Pd[ci] - p_others) / nslack;
      gp = (float)p; gq = (float)q;
      gv = (float)(c.vm[ci] * g.sub_vn_kv[g.gen_sub[i]]);
      gth = (float)(c.va[ci] * RAD2DEG);
    }
    if (have) { out[oo.gen_p + i] = gp; out[oo.gen_q + i] = gq; out[oo.gen_v + i] = gv; out[oo.gen_th + i] = gth; }
  }
  int* to = b.topo_out + (size_t)inst * g.dim_topo;
  for (int i = tid; i < g.dim_topo; i += WAVE) { const int v = topo_g[i]; to[i] = v >= 1 ? v : -1; }
  __syncthreads();
  for (int l = tid; l < g.n_line; l += WAVE) {
    if (c.lor_c[l] < 0) { to[g.line_or_pos[l]] = -1; to[g.line_ex_pos[l]] = -1; }
  }
  double* bvm = b.bus_vm + (size_t)inst * g.nb_tot;
  double* bva = b.bus_va + (size_t)inst * g.nb_tot;
  const double nand = __builtin_nan("");
  for (int i = tid; i < g.nb_tot; i += WAVE) {
    const int ci = c.gmap[i];
    bvm[i] = ci >= 0 ? c.vm[ci] : nand;
    bva[i] = ci >= 0 ? c.va[ci] * RAD2DEG : nand;
  }
  GPF_STAMP(7);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
template <int NMAX, int LPR>
__global__ __launch_bounds__(WAVE, GPF_MINW(NMAX)) void runpf_small_kernel(const DevParams* __restrict__ P, int lane0, int nbc,
                                                                           int nrows, int is_dc, int max_iter, double tol_pu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int inst = lane0 + blockIdx.x;
  const int tid = threadIdx.x;
  CarveS c;
  carve_small<NMAX, LPR>(c, smem, P->g, nbc, nrows);
  int n_iter, nb;
  const int st = solve_instance_small<NMAX, LPR>(P, c, inst, nbc, nrows, is_dc, max_iter, tol_pu, tid, false, n_iter, nb);
  __syncthreads();
  if (st != 0) write_nan_results(P->g, P->b, inst, tid);
  if (tid == 0) {
    int* s = P->b.status + (size_t)inst * 4;
    s[0] = st; s[1] = n_iter; s[2] = nb; s[3] = 0;
  }
}

template <int NMAX, int LPR>
__global__ __launch_bounds__(WAVE, GPF_MINW(NMAX)) void step_small_kernel(const DevParams* __restrict__ P, int nbc, int nrows,
                                                                          int max_iter, double tol_pu, StepArgs sa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const GridDev& g = P->g;
  const Bufs& b = P->b;
  const OutOff& oo = P->oo;
  const int inst = blockIdx.x;
  const int tid = threadIdx.x;
  CarveS c;
  carve_small<NMAX, LPR>(c, smem, g, nbc, nrows);
  GPF_STAMP(8);
  // ---- K9: chronics row -> injections (kept in LDS for the solver, written back to the lane state in HBM) ---------------------
  {
    const int tab = b.lane_table ? b.lane_table[inst] : 0;
    const int off = b.lane_offset ? b.lane_offset[inst] : 0;
    int row = (sa.t + off) % sa.T;
    if (row < 0) row += sa.T;
    const float* __restrict__ ch = b.chron + ((size_t)tab * sa.T + row) * g.n_chron;
    const float* __restrict__ sc = b.lane_scale ? b.lane_scale + (size_t)inst * 2 * g.n_load : nullptr;
    double* inj_g = b.inj + (size_t)inst * g.n_inj;
    // storage / shunt set-points are not driven by the chronics: stage the lane's current values
    for (int i = oo.inj_sto_p + tid; i < g.n_inj; i += WAVE) c.inj[i] = inj_g[i];
    double sum_load = 0.0, sum_prod = 0.0;
    for (int i = tid; i < g.n_load; i += WAVE) {
      float lp = ch[i], lq = ch[g.n_load + i];
      if (sc) { lp *= sc[i]; lq *= sc[g.n_load + i]; }
      c.inj[oo.inj_load_p + i] = (double)lp;
      c.inj[oo.inj_load_q + i] = (double)lq;
      inj_g[oo.inj_load_p + i] = (double)lp;
      inj_g[oo.inj_load_q + i] = (double)lq;
      sum_load += (double)lp;
    }
    for (int i = tid; i < g.n_gen; i += WAVE)
      if (!g.gen_slack[i]) sum_prod += (double)ch[2 * g.n_load + i];
    float scale_p = 1.0f;
    if (sa.rebalance_on) {
      sum_load = wave_sum(sum_load);
      sum_prod = wave_sum(sum_prod);
      scale_p = (sum_prod > 0.0) ? (float)(sa.rebalance * sum_load / sum_prod) : 1.0f;
    }
    for (int i = tid; i < g.n_gen; i += WAVE) {
      float pp = ch[2 * g.n_load + i];
      if (!g.gen_slack[i]) pp *= scale_p;
      const float pv_kv = ch[2 * g.n_load + g.n_gen + i];
      const float vn = (float)g.sub_vn_kv[g.gen_sub[i]];
      const double vm_pu = (double)(pv_kv / vn);         // float32 division, as pandaPowerBackend.py:927
      c.inj[oo.inj_gen_p + i] = (double)pp;
      c.inj[oo.inj_gen_vm + i] = vm_pu;
      inj_g[oo.inj_gen_p + i] = (double)pp;
      inj_g[oo.inj_gen_vm + i] = vm_pu;
    }
    __syncthreads();
  }
  int n_iter = 0, nb = 0, st = 0, rounds = 0;
  int* ovc = b.overflow_count + (size_t)inst * g.n_line;
  int* dround = b.disc_round + (size_t)inst * g.n_line;
  float* rho = b.rho + (size_t)inst * g.n_line;
  float* out = b.out + (size_t)inst * g.n_out;
  int* topo = b.topo + (size_t)inst * g.dim_topo;
  constexpr int MAXK = 4;
  int loc[MAXK];
  bool inc[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int l = tid + k * WAVE;
    loc[k] = (l < g.n_line) ? ovc[l] : 0;
    inc[k] = false;
    if (l < g.n_line) dround[l] = -1;
  }
  while (true) {
    st = solve_instance_small<NMAX, LPR>(P, c, inst, nbc, nrows, sa.is_dc, max_iter, tol_pu, tid, true, n_iter, nb);
    __syncthreads();
    if (st != 0 || !sa.cascade || rounds >= sa.max_rounds) break;   // at most max_rounds re-solves
    int any_disc = 0;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int l = tid + k * WAVE;
      if (l >= g.n_line) continue;
      const float a = out[oo.a_or + l];
      const float lim = b.thermal_limit[l];
      const bool on = c.lor_c[l] >= 0;
      bool disc = on && (a > sa.hard_overflow * lim);
      if (on && (a > sa.soft_overflow * lim) && !inc[k]) { loc[k] += 1; inc[k] = true; }
      if (on && loc[k] > sa.nb_ts_allowed) disc = true;
      if (disc) {
        topo[g.line_or_pos[l]] = -1;
        topo[g.line_ex_pos[l]] = -1;
        dround[l] = rounds;
        any_disc = 1;
      }
    }
    __syncthreads();
    if (!__any(any_disc)) break;
    ++rounds;
  }
  GPF_STAMP(9);
  if (st != 0) write_nan_results(g, b, inst, tid);
  __syncthreads();
  for (int l = tid; l < g.n_line; l += WAVE) {
    const float lim = b.thermal_limit[l];
    const float a = out[oo.a_or + l];
    rho[l] = a / lim;
    if (a > sa.soft_overflow * lim) ovc[l] += 1; else ovc[l] = 0;
  }
  if (tid == 0) {
    int* s = b.status + (size_t)inst * 4;
    s[0] = st; s[1] = n_iter; s[2] = nb; s[3] = rounds;
  }
}

}  // namespace gpf
