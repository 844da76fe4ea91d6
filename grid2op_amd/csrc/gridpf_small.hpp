// gridpf_small.hpp -- kernel v2 for small / medium grids (Newton unknowns n <= NMAX <= 64).
//
// Same pipeline and same arithmetic as gridpf_kernels.hpp (one wavefront per grid instance), but the
// linear algebra is REGISTER RESIDENT:
//   * each SIMD lane owns ONE ROW of the compact Jacobian [J | -F] in VGPRs (statically indexed, fully unrolled);
//   * Gauss-Jordan elimination with partial pivoting: the pivot is found with a 6-step DPP max-reduction on a
//     32-bit key (high word of |a_ik| with the lane id in the low 6 bits), the pivot row is broadcast with
//     v_readlane (wave-uniform lane -> SGPR operands of the FMAs); no LDS traffic, no barrier, no back-substitution;
//   * the Jacobian rows are assembled by the owning lane (dense over the buses, every entry written once) through
//     a conflict-free LDS row buffer (row stride = NMAX+2 doubles) so that the register file is indexed statically;
//   * sincos is a branch-free Cody-Waite + fdlibm-kernel implementation (angles here are a few radians at most);
//   * kernel parameters are one pointer to a device-resident DevParams block (no SGPR spilling).
#pragma once
#include "gridpf_kernels.hpp"

namespace gpf {

struct DevParams {
  GridDev g;
  Bufs b;
  OutOff oo;
};

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

// sin / cos for |x| up to a few thousand radians: Cody-Waite reduction by pi/2 (3-part constant with FMA) +
// the fdlibm __kernel_sin / __kernel_cos minimax polynomials on [-pi/4, pi/4] (< 1 ulp).
__device__ __forceinline__ void fast_sincos(double x, double& s, double& c) {
  const double TWO_OVER_PI = 0.63661977236758134308;
  const double P1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double P2 = 6.07710050650619224932e-11;   // next 33 bits
  const double P3 = 2.02226624879595063154e-21;   // tail
  const double kf = rint(x * TWO_OVER_PI);
  double r = fma(-kf, P1, x);
  r = fma(-kf, P2, r);
  r = fma(-kf, P3, r);
  const int q = (int)kf;
  const double z = r * r;
  // sin kernel
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
  const double sr = fma(r * z, ps, r);
  // cos kernel
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
// Register-resident Gauss-Jordan.  Lane `lane` owns row `lane` (rows >= n are idle).  a[0..NMAX-1] = matrix row,
// a[NMAX] = right-hand side.  Columns k >= n are skipped.  On return the lane that pivoted column k holds x_k in
// `x` and k in `mycol` (-1 for idle lanes).  Returns false on a zero pivot.
template <int NMAX>
__device__ __forceinline__ bool gj_solve(double (&a)[NMAX + 1], int n, int lane, double& x, int& mycol) {
  bool used = lane >= n;
  bool ok = true;
  double piv = 1.0;
  mycol = -1;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k < n) {   // wave-uniform
      const double av = fabs(a[k]);
      unsigned key = (((unsigned)__double2hiint(av)) & ~63u) | (unsigned)lane;
      if (used) key = (unsigned)lane;
      const unsigned best = wave_umax(key);
      const int p = (int)(best & 63u);
      if ((best >> 6) == 0u) ok = false;
      const double pk = readlane_f64(a[k], p);
      const double rp = 1.0 / pk;
      double m = a[k] * rp;
      if (lane == p) { used = true; mycol = k; m = 0.0; piv = pk; }
#pragma unroll
      for (int j = k + 1; j <= NMAX; ++j) {
        const double pj = readlane_f64(a[j], p);
        a[j] = fma(-m, pj, a[j]);
      }
    }
  }
  x = a[NMAX] / piv;
  return ok;
}

// LDS carve of the small kernels ------------------------------------------------------------------------------------
struct CarveS {
  double *vm, *va, *e, *f, *ivm, *Psp, *Qsp, *Sre, *Sim, *vset, *Pd, *Qd, *Gs, *dx;
  double* Y;      // [nbc][ldy] complex (re, im interleaved), ldy = nbc | 1
  double* R;      // [nrows][NMAX + 2] row buffer (Jacobian / B' rows); Sre/Sim alias its head after the Newton loop
  int *gmap, *gid, *btype, *pidx, *qidx, *lab, *pbus, *qbus;
  int *lor_c, *lex_c, *gen_c, *load_c, *sto_c, *sh_c;
};

template <int NMAX>
__host__ __device__ inline size_t lds_bytes_small(const GridDev& g, int nbc, int nrows) {
  const size_t ldy = (size_t)(nbc | 1);
  size_t rbuf = (size_t)nrows * (NMAX + 2);
  if (rbuf < (size_t)2 * nbc) rbuf = (size_t)2 * nbc;
  size_t nd = (size_t)11 * nbc + (size_t)NMAX + 2 * (size_t)nbc * ldy + rbuf;
  size_t ni = (size_t)g.nb_tot + 4 * (size_t)nbc + 2 * (size_t)NMAX + 2 * (size_t)g.n_line + g.n_gen + g.n_load + g.n_sto +
              g.n_shunt;
  return nd * 8 + ((ni * 4 + 7) & ~(size_t)7);
}

template <int NMAX>
__device__ inline void carve_small(CarveS& c, unsigned char* base, const GridDev& g, int nbc, int nrows) {
  const int ldy = nbc | 1;
  double* d = reinterpret_cast<double*>(base);
  size_t rbuf = (size_t)nrows * (NMAX + 2);
  if (rbuf < (size_t)2 * nbc) rbuf = (size_t)2 * nbc;
  c.R = d; c.Sre = d; c.Sim = d + nbc; d += rbuf;     // first: 16-byte aligned rows for ds_read_b128
  c.Y = d; d += (size_t)2 * nbc * ldy;
  c.vm = d; d += nbc; c.va = d; d += nbc; c.e = d; d += nbc; c.f = d; d += nbc; c.ivm = d; d += nbc;
  c.Psp = d; d += nbc; c.Qsp = d; d += nbc;
  c.vset = d; d += nbc; c.Pd = d; d += nbc; c.Qd = d; d += nbc; c.Gs = d; d += nbc;
  c.dx = d; d += NMAX;
  int* i = reinterpret_cast<int*>(d);
  c.gmap = i; i += g.nb_tot;
  c.gid = nullptr; c.btype = i; i += nbc; c.pidx = i; i += nbc; c.qidx = i; i += nbc; c.lab = i; i += nbc;
  c.pbus = i; i += NMAX; c.qbus = i; i += NMAX;
  c.lor_c = i; i += g.n_line; c.lex_c = i; i += g.n_line;
  c.gen_c = i; i += g.n_gen; c.load_c = i; i += g.n_load; c.sto_c = i; i += g.n_sto; c.sh_c = i; i += g.n_shunt;
}

// K1 for the small kernels: identical to build_topology() but on CarveS (kept separate to avoid a template on the
// carve type in the generic kernels).
__device__ inline int build_topology_s(const GridDev& g, CarveS& c, const int* __restrict__ topo,
                                       const int* __restrict__ shunt_bus, unsigned char* __restrict__ status_out, int nbc,
                                       int tid) {
  Carve cc{};
  cc.gmap = c.gmap; cc.gid = c.lab;  /* gid unused by the small kernels: lab doubles as scratch */ cc.lor_c = c.lor_c; cc.lex_c = c.lex_c; cc.gen_c = c.gen_c; cc.load_c = c.load_c;
  cc.sto_c = c.sto_c; cc.sh_c = c.sh_c;
  return build_topology(g, cc, topo, shunt_bus, status_out, nbc, tid);
}

// ---------------------------------------------------------------------------------------------------
template <int NMAX>
__device__ inline int solve_instance_small(const DevParams* __restrict__ P, CarveS& c, int inst, int nbc, int nrows, int is_dc,
                                           int max_iter, double tol_pu, int tid, int& n_iter_out, int& nb_out) {
  const GridDev& g = P->g;
  const Bufs& b = P->b;
  const OutOff& oo = P->oo;
  const double* __restrict__ inj = b.inj + (size_t)inst * g.n_inj;
  const int* __restrict__ topo = b.topo + (size_t)inst * g.dim_topo;
  const int* __restrict__ shb = b.shunt_bus + (size_t)inst * g.n_shunt;
  unsigned char* lstat = b.line_status + (size_t)inst * g.n_line;
  constexpr int LDR = NMAX + 2;
  n_iter_out = 0;
  nb_out = 0;

  // ---- K1 ----------------------------------------------------------------------------------------------------------
  const int nb = build_topology_s(g, c, topo, shb, lstat, nbc, tid);
  if (nb < 0) return 5;
  nb_out = nb;
  const int ldy = nbc | 1;
  const double sn = g.sn_mva;
  const double inv_sn = 1.0 / sn;

  // ---- bus types / injections: thread per bus gathers over the elements (oracle summation order) ---------------------
  for (int ci = tid; ci < nb; ci += WAVE) {
    int bt = BT_PQ;
    double Pg = 0.0, vs = 1.0;
    for (int i = 0; i < g.n_gen; ++i) {
      if (c.gen_c[i] == ci) {
        if (g.gen_slack[i]) bt = BT_REF;
        else { if (bt != BT_REF) bt = BT_PV; Pg += inj[oo.inj_gen_p + i] * inv_sn; }
        vs = inj[oo.inj_gen_vm + i];
      }
    }
    double pd = 0.0, qd = 0.0;
    for (int i = 0; i < g.n_load; ++i)
      if (c.load_c[i] == ci) { pd += inj[oo.inj_load_p + i]; qd += inj[oo.inj_load_q + i]; }
    for (int i = 0; i < g.n_sto; ++i)
      if (c.sto_c[i] == ci) { pd += inj[oo.inj_sto_p + i]; qd += inj[oo.inj_sto_q + i]; }
    double gs = 0.0;
    for (int i = 0; i < g.n_shunt; ++i)
      if (c.sh_c[i] == ci) gs += inj[oo.inj_sh_p + i] * g.shunt_fact[i] * inv_sn;
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
      c.pidx[ci] = pi;
      c.qidx[ci] = qi;
      if (pi >= 0 && pi < NMAX) c.pbus[pi] = ci;
      if (qi >= 0 && qi < NMAX) c.qbus[qi] = ci;
    }
    npvpq += __popcll(mp);
    npq += __popcll(mq);
    nref += __popcll(mr);
  }
  __syncthreads();
  if (nref == 0) return 3;
  const int n = npvpq + npq;
  if (n > NMAX || n > nrows || npvpq > nrows) return 5;

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

  // ---- K2 + K3 assembly: one pass over the branches per bus row builds the Ybus row (AC) and the B' row (DC) ---------------
  // B' rows live in the row buffer R (row = pidx of the bus), zeroed by the owning lane first.
  for (int ci = tid; ci < nb; ci += WAVE) {
    double* Yr = c.Y + (size_t)2 * ci * ldy;
    for (int j = 0; j < 2 * nb; ++j) Yr[j] = 0.0;
    const int pi = c.pidx[ci];
    double* Rr = c.R + (size_t)(pi >= 0 ? pi : 0) * LDR;
    if (pi >= 0)
      for (int j = 0; j < npvpq; ++j) Rr[j] = 0.0;
    double diag = 0.0;
    for (int l = 0; l < g.n_line; ++l) {
      const int f = c.lor_c[l], t = c.lex_c[l];
      if (f < 0 || (f != ci && t != ci)) continue;
      const double* y = g.br_y + (size_t)8 * l;
      if (f == ci) {
        Yr[2 * f] += y[0]; Yr[2 * f + 1] += y[1];
        Yr[2 * t] += y[2]; Yr[2 * t + 1] += y[3];
      }
      if (t == ci) {
        Yr[2 * f] += y[4]; Yr[2 * f + 1] += y[5];
        Yr[2 * t] += y[6]; Yr[2 * t + 1] += y[7];
      }
      if (pi >= 0 && f != t) {
        const double bb = g.br_bdc[l];
        diag += bb;
        const int po = c.pidx[(f == ci) ? t : f];
        if (po >= 0) Rr[po] -= bb;
      }
    }
    for (int s = 0; s < g.n_shunt; ++s) {
      if (c.sh_c[s] == ci) {
        Yr[2 * ci] += inj[oo.inj_sh_p + s] * g.shunt_fact[s] * inv_sn;
        Yr[2 * ci + 1] -= inj[oo.inj_sh_q + s] * g.shunt_fact[s] * inv_sn;
      }
    }
    if (pi >= 0) {
      Rr[pi] += diag;
      Rr[NMAX] = c.Psp[ci] - c.Gs[ci];
    }
  }
  __syncthreads();

  double a[NMAX + 1];
  // ---- K3: DC solve ------------------------------------------------------------------------------------------------------
  {
    const double* Rr = c.R + (size_t)(tid < nrows ? tid : 0) * LDR;
#pragma unroll
    for (int j = 0; j <= NMAX; ++j) a[j] = (tid < npvpq && (j < npvpq || j == NMAX)) ? Rr[j] : 0.0;
    double x;
    int mycol;
    bool ok = gj_solve<NMAX>(a, npvpq, tid, x, mycol);
    if (mycol >= 0) {
      c.dx[mycol] = x;
      if (!(fabs(x) < 1e300)) ok = false;
    }
    __syncthreads();
    if (__any(!ok)) return 4;
    for (int ci = tid; ci < nb; ci += WAVE) {
      const int pi = c.pidx[ci];
      c.va[ci] = (pi >= 0) ? c.dx[pi] : 0.0;
      c.vm[ci] = (c.btype[ci] == BT_PQ) ? 1.0 : c.vset[ci];
    }
    __syncthreads();
  }

  int status = 0;
  int it = 0;
  if (!is_dc) {
    // ---- K4/K5: Newton-Raphson ---------------------------------------------------------------------------------------------
    bool converged = false;
    const bool row_on = tid < n;
    const bool isQ = tid >= npvpq;
    const int ib = row_on ? (isQ ? c.qbus[tid - npvpq] : c.pbus[tid]) : 0;
    const double Pi = c.Psp[ib], Qi = c.Qsp[ib];
    const int pii = c.pidx[ib], qii = c.qidx[ib];
    double* Rr = c.R + (size_t)(tid < nrows ? tid : 0) * LDR;
    double Sr_last = 0.0, Si_last = 0.0;
    while (true) {
      for (int ci = tid; ci < nb; ci += WAVE) {
        double s, co;
        fast_sincos(c.va[ci], s, co);
        const double vmi = c.vm[ci];
        c.e[ci] = vmi * co;
        c.f[ci] = vmi * s;
        c.ivm[ci] = 1.0 / vmi;
      }
      __syncthreads();
      // Row assembly (dense over the buses): T_ij = V_i conj(Y_ij V_j); S_i = sum_j T_ij
      double fabs_mis = 0.0;
      bool bad = false;
      if (row_on) {
        const double ei = c.e[ib], fi = c.f[ib];
        const double* Yr = c.Y + (size_t)2 * ib * ldy;
        double Sr = 0.0, Si = 0.0, Tr = 0.0, Ti = 0.0;
        for (int j = 0; j < nb; ++j) {
          const double yr = Yr[2 * j], yi = Yr[2 * j + 1];
          const double ej = c.e[j], fj = c.f[j];
          const double aa = yr * ej - yi * fj, bb = yr * fj + yi * ej;
          const double tr_ = ei * aa + fi * bb;
          const double ti_ = fi * aa - ei * bb;
          Sr += tr_;
          Si += ti_;
          const int pj = c.pidx[j], qj = c.qidx[j];
          if (pj >= 0) Rr[pj] = isQ ? -tr_ : ti_;
          if (qj >= 0) Rr[npvpq + qj] = (isQ ? ti_ : tr_) * c.ivm[j];
          if (j == ib) { Tr = tr_; Ti = ti_; }
        }
        const double ivmi = c.ivm[ib];
        // diagonal block: dS/dVa_ii = j (S - T_ii), dS/dVm_ii = (T_ii + S) / |V_i|
        Rr[pii] = isQ ? (Sr - Tr) : (Ti - Si);
        if (qii >= 0) Rr[npvpq + qii] = (isQ ? (Ti + Si) : (Tr + Sr)) * ivmi;
        const double mis = isQ ? (Si - Qi) : (Sr - Pi);
        Rr[NMAX] = -mis;
        fabs_mis = fabs(mis);
        if (!(fabs_mis <= 1e300)) bad = true;
        Sr_last = Sr;
        Si_last = Si;
      }
      // reference buses (no row): their S is needed by the result stage only -> computed there
      // ||F||inf < tol  <=>  no row has |F_row| >= tol (no floating-point reduction needed)
      const bool any_ge = __any(row_on && !(fabs_mis < tol_pu));
      if (__any(bad)) { status = 1; break; }
      if (!any_ge) { converged = true; break; }
      if (it >= max_iter) break;
      ++it;
#pragma unroll
      for (int j = 0; j <= NMAX; ++j) a[j] = (row_on && (j < n || j == NMAX)) ? Rr[j] : 0.0;
      double x;
      int mycol;
      bool ok = gj_solve<NMAX>(a, n, tid, x, mycol);
      if (mycol >= 0) {
        c.dx[mycol] = x;
        if (!(fabs(x) < 1e300)) ok = false;
      }
      __syncthreads();
      if (__any(!ok)) { status = 4; break; }
      for (int ci = tid; ci < nb; ci += WAVE) {
        const int pi = c.pidx[ci], qi = c.qidx[ci];
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
    __syncthreads();                       // the row buffer is dead from here on: its head becomes Sre / Sim
    if (row_on && !isQ) { c.Sre[ib] = Sr_last; c.Sim[ib] = Si_last; }
  }
  n_iter_out = it;
  if (status != 0) return status;

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
      if (c.pidx[ci] >= 0) continue;
      const double* Yr = c.Y + (size_t)2 * ci * ldy;
      double ir = 0.0, ii = 0.0;
      for (int j = 0; j < nb; ++j) {
        const double yr = Yr[2 * j], yi = Yr[2 * j + 1];
        ir += yr * c.e[j] - yi * c.f[j];
        ii += yr * c.f[j] + yi * c.e[j];
      }
      c.Sre[ci] = c.e[ci] * ir + c.f[ci] * ii;
      c.Sim[ci] = c.f[ci] * ir - c.e[ci] * ii;
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
        const double* y = g.br_y + (size_t)8 * l;
        const double ef = c.e[f], ff = c.f[f], et = c.e[t], ft = c.f[t];
        const double ifr = y[0] * ef - y[1] * ff + y[2] * et - y[3] * ft;
        const double ifi = y[0] * ff + y[1] * ef + y[2] * ft + y[3] * et;
        const double itr = y[4] * ef - y[5] * ff + y[6] * et - y[7] * ft;
        const double iti = y[4] * ff + y[5] * ef + y[6] * ft + y[7] * et;
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
  for (int i = tid; i < g.n_gen; i += WAVE) {
    const int ci = c.gen_c[i];
    float gp = 0.f, gq = 0.f, gv = 0.f, gth = 0.f;
    if (ci >= 0) {
      int cnt = 0, nslack = 0;
      double qmin_t = 0.0, qmax_t = 0.0, p_others = 0.0;
      for (int k = 0; k < g.n_gen; ++k) {
        if (c.gen_c[k] == ci) {
          ++cnt;
          qmin_t += g.gen_min_q[k];
          qmax_t += g.gen_max_q[k];
          if (g.gen_slack[k]) ++nslack; else p_others += inj[oo.inj_gen_p + k];
        }
      }
      const double qtot = c.Sim[ci] * sn + c.Qd[ci];
      double q;
      if (is_dc) q = 0.0;
      else if (cnt == 1) q = qtot;
      else if (qmin_t == qmax_t) q = qtot / cnt;
      else q = g.gen_min_q[i] + (qtot - qmin_t) / (qmax_t - qmin_t + 2.220446049250313e-16) * (g.gen_max_q[i] - g.gen_min_q[i]);
      double p = inj[oo.inj_gen_p + i];
      if (g.gen_slack[i]) p = (c.Sre[ci] * sn + c.Pd[ci] - p_others) / nslack;
      gp = (float)p; gq = (float)q;
      gv = (float)(c.vm[ci] * g.sub_vn_kv[g.gen_sub[i]]);
      gth = (float)(c.va[ci] * RAD2DEG);
    }
    out[oo.gen_p + i] = gp; out[oo.gen_q + i] = gq; out[oo.gen_v + i] = gv; out[oo.gen_th + i] = gth;
  }
  int* to = b.topo_out + (size_t)inst * g.dim_topo;
  for (int i = tid; i < g.dim_topo; i += WAVE) { const int v = topo[i]; to[i] = v >= 1 ? v : -1; }
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
  return 0;
}

// ---------------------------------------------------------------------------------------------------
template <int NMAX>
__global__ __launch_bounds__(WAVE, (NMAX <= 24 ? 4 : NMAX <= 48 ? 3 : 2)) void runpf_small_kernel(const DevParams* __restrict__ P, int lane0, int nbc, int nrows, int is_dc,
                                                           int max_iter, double tol_pu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int inst = lane0 + blockIdx.x;
  const int tid = threadIdx.x;
  CarveS c;
  carve_small<NMAX>(c, smem, P->g, nbc, nrows);
  int n_iter, nb;
  const int st = solve_instance_small<NMAX>(P, c, inst, nbc, nrows, is_dc, max_iter, tol_pu, tid, n_iter, nb);
  __syncthreads();
  if (st != 0) write_nan_results(P->g, P->b, inst, tid);
  if (tid == 0) {
    int* s = P->b.status + (size_t)inst * 4;
    s[0] = st; s[1] = n_iter; s[2] = nb; s[3] = 0;
  }
}

template <int NMAX>
__global__ __launch_bounds__(WAVE, (NMAX <= 24 ? 4 : NMAX <= 48 ? 3 : 2)) void step_small_kernel(const DevParams* __restrict__ P, int nbc, int nrows, int max_iter,
                                                          double tol_pu, StepArgs sa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const GridDev& g = P->g;
  const Bufs& b = P->b;
  const OutOff& oo = P->oo;
  const int inst = blockIdx.x;
  const int tid = threadIdx.x;
  CarveS c;
  carve_small<NMAX>(c, smem, g, nbc, nrows);
  // ---- K9: chronics row -> injections ---------------------------------------------------------------------------------------
  {
    const int tab = b.lane_table ? b.lane_table[inst] : 0;
    const int off = b.lane_offset ? b.lane_offset[inst] : 0;
    int row = (sa.t + off) % sa.T;
    if (row < 0) row += sa.T;
    const float* __restrict__ ch = b.chron + ((size_t)tab * sa.T + row) * g.n_chron;
    const float* __restrict__ sc = b.lane_scale ? b.lane_scale + (size_t)inst * 2 * g.n_load : nullptr;
    double* inj = b.inj + (size_t)inst * g.n_inj;
    double sum_load = 0.0, sum_prod = 0.0;
    for (int i = tid; i < g.n_load; i += WAVE) {
      float lp = ch[i], lq = ch[g.n_load + i];
      if (sc) { lp *= sc[i]; lq *= sc[g.n_load + i]; }
      inj[oo.inj_load_p + i] = (double)lp;
      inj[oo.inj_load_q + i] = (double)lq;
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
      inj[oo.inj_gen_p + i] = (double)pp;
      inj[oo.inj_gen_vm + i] = (double)(pv_kv / vn);
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
    st = solve_instance_small<NMAX>(P, c, inst, nbc, nrows, 0, max_iter, tol_pu, tid, n_iter, nb);
    __syncthreads();
    if (st != 0 || !sa.cascade) break;
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
    if (rounds >= sa.max_rounds) break;
    ++rounds;
  }
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
