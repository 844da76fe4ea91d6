/* pf_oracle.c -- CPU ORACLE (plain C99, float64).  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build / load this file.
 * It restates, for one grid instance at a time, the arithmetic the reference delegates to the
 * third-party package pandapower (pandapower>=3.1.1, pyproject.toml:15 of the reference; not vendored,
 * not installable here) from grid2op/Backend/pandaPowerBackend.py:
 *   - pp.runpp(check_connectivity=False, init="dc", max_iteration=10)      (:1097-1105)
 *   - pp.rundcpp(check_connectivity=True, init="flat")                      (:1090)
 *   - the read-back of _fetch_data_pf_converged / _gens_info / ...          (:1122-1218, 1526-1647)
 * following pandapower's published pd2ppc -> makeYbus -> newtonpf -> pfsoln chain (formulas:
 * SURVEY.md section 8 row A4').  It mirrors oracle/pf_oracle.py (numpy), which is the version pinned to
 * the reference's golden vectors; tests/test_oracle_c.py checks the two against each other and against
 * the same golden vectors.  Dense algebra throughout (Gaussian elimination with partial pivoting).
 *
 * The grid description struct and the result-row layout are those of include/gridpf.h, so that the
 * parity tests can diff rows directly.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gridpf.h"

typedef double complex cplx;

static int dense_solve(double* A, double* b, int n) {
  /* in-place LU with partial pivoting; returns 0 on success */
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      double v = fabs(A[i * n + k]);
      if (v > best) { best = v; p = i; }
    }
    if (!(best > 1e-300) || !(best < 1e300)) return 1;
    if (p != k) {
      for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
      double t = b[k]; b[k] = b[p]; b[p] = t;
    }
    double rp = 1.0 / A[k * n + k];
    for (int i = k + 1; i < n; ++i) {
      double m = A[i * n + k] * rp;
      if (m == 0.0) continue;
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= m * A[k * n + j];
      b[i] -= m * b[k];
    }
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * b[j];
    b[k] = s / A[k * n + k];
  }
  return 0;
}

typedef struct {
  int p_or, q_or, v_or, a_or, th_or, p_ex, q_ex, v_ex, a_ex, th_ex;
  int gen_p, gen_q, gen_v, gen_th, load_p, load_q, load_v, load_th, sto_p, sto_q, sto_v, sto_th, sh_p, sh_q, sh_v;
  int n_out;
  int inj_gen_p, inj_gen_vm, inj_load_p, inj_load_q, inj_sto_p, inj_sto_q, inj_sh_p, inj_sh_q, n_inj;
} offsets_t;

static offsets_t make_offsets(const gpf_grid_desc* d) {
  offsets_t o;
  int nl = d->n_line, ng = d->n_gen, nd = d->n_load, ns = d->n_storage, nsh = d->n_shunt, k = 0;
  o.p_or = k; k += nl; o.q_or = k; k += nl; o.v_or = k; k += nl; o.a_or = k; k += nl; o.th_or = k; k += nl;
  o.p_ex = k; k += nl; o.q_ex = k; k += nl; o.v_ex = k; k += nl; o.a_ex = k; k += nl; o.th_ex = k; k += nl;
  o.gen_p = k; k += ng; o.gen_q = k; k += ng; o.gen_v = k; k += ng; o.gen_th = k; k += ng;
  o.load_p = k; k += nd; o.load_q = k; k += nd; o.load_v = k; k += nd; o.load_th = k; k += nd;
  o.sto_p = k; k += ns; o.sto_q = k; k += ns; o.sto_v = k; k += ns; o.sto_th = k; k += ns;
  o.sh_p = k; k += nsh; o.sh_q = k; k += nsh; o.sh_v = k; k += nsh;
  o.n_out = k;
  k = 0;
  o.inj_gen_p = k; k += ng; o.inj_gen_vm = k; k += ng; o.inj_load_p = k; k += nd; o.inj_load_q = k; k += nd;
  o.inj_sto_p = k; k += ns; o.inj_sto_q = k; k += ns; o.inj_sh_p = k; k += nsh; o.inj_sh_q = k; k += nsh;
  o.n_inj = k;
  return o;
}

int pfo_n_out(const gpf_grid_desc* d) { return make_offsets(d).n_out; }
int pfo_n_inj(const gpf_grid_desc* d) { return make_offsets(d).n_inj; }

static void fill_fail(const gpf_grid_desc* d, const offsets_t* o, double* out, int32_t* topo_out, int32_t* shunt_bus_out,
                      uint8_t* line_status, double* bus_vm, double* bus_va) {
  int nbt = d->n_sub * d->n_busbar;
  for (int i = 0; i < o->n_out; ++i) out[i] = NAN;
  for (int i = 0; i < d->dim_topo; ++i) topo_out[i] = -1;
  for (int i = 0; i < d->n_shunt; ++i) shunt_bus_out[i] = -1;
  for (int i = 0; i < d->n_line; ++i) line_status[i] = 0;
  for (int i = 0; i < nbt; ++i) { bus_vm[i] = NAN; bus_va[i] = NAN; }
}

/* One power flow of one grid instance.  status4 = {GPF_ST_*, n_iter, n_active_bus, 0}. */
int pfo_solve(const gpf_grid_desc* d, const double* inj, const int32_t* topo, const int32_t* shunt_bus, int is_dc,
              int max_iter, double tol_mva, double* out, int32_t* topo_out, int32_t* shunt_bus_out, uint8_t* line_status,
              int32_t* status4, double* bus_vm, double* bus_va) {
  const offsets_t o = make_offsets(d);
  const int ns_ = d->n_sub, nbt = d->n_sub * d->n_busbar;
  const int nl = d->n_line, ng = d->n_gen, nd = d->n_load, nst = d->n_storage, nsh = d->n_shunt;
  const double sn = d->sn_mva;
  int rc = GPF_ST_CONVERGED, n_iter = 0, nb = 0;

  int* lor = (int*)malloc(sizeof(int) * (size_t)(2 * nl + ng + nd + nst + nsh + 4 * nbt + 8));
  int* lex = lor + nl;
  int* gbus = lex + nl;
  int* lbus = gbus + ng;
  int* sbus = lbus + nd;
  int* shb = sbus + nst;
  int* active = shb + nsh;
  int* btype = active + nbt;   /* 0 PQ, 1 PV, 2 REF */
  int* pidx = btype + nbt;
  int* qidx = pidx + nbt;
  double* P = (double*)calloc((size_t)(8 * nbt), sizeof(double));
  double* Q = P + nbt;
  double* Pd = Q + nbt;
  double* Qd = Pd + nbt;
  double* Gs = Qd + nbt;
  double* vset = Gs + nbt;
  double* va = vset + nbt;
  double* vm = va + nbt;
  cplx* Y = NULL;
  cplx* V = NULL;
  double* J = NULL;
  double* F = NULL;
  double* B = NULL;

  /* element buses, line status (a line is in service iff both ends are connected) */
  memset(active, 0, sizeof(int) * (size_t)nbt);
  for (int l = 0; l < nl; ++l) {
    int bo = topo[d->line_or_pos_topo_vect[l]], be = topo[d->line_ex_pos_topo_vect[l]];
    int on = (bo >= 1 && be >= 1);
    line_status[l] = (uint8_t)on;
    lor[l] = on ? d->line_or_sub[l] + (bo - 1) * ns_ : -1;
    lex[l] = on ? d->line_ex_sub[l] + (be - 1) * ns_ : -1;
    if (on) { active[lor[l]] = 1; active[lex[l]] = 1; }
  }
#define ELBUS(arr, n, sub, posarr)                                \
  for (int i = 0; i < (n); ++i) {                                 \
    int b_ = topo[(posarr)[i]];                                   \
    arr[i] = b_ >= 1 ? (sub)[i] + (b_ - 1) * ns_ : -1;            \
    if (arr[i] >= 0) active[arr[i]] = 1;                          \
  }
  ELBUS(gbus, ng, d->gen_sub, d->gen_pos_topo_vect)
  ELBUS(lbus, nd, d->load_sub, d->load_pos_topo_vect)
  ELBUS(sbus, nst, d->storage_sub, d->storage_pos_topo_vect)
#undef ELBUS
  for (int i = 0; i < nsh; ++i) {
    int b_ = shunt_bus[i];
    shb[i] = b_ >= 1 ? d->shunt_sub[i] + (b_ - 1) * ns_ : -1;
    if (shb[i] >= 0) active[shb[i]] = 1;
  }
  for (int i = 0; i < d->dim_topo; ++i) topo_out[i] = topo[i] >= 1 ? topo[i] : -1;
  for (int l = 0; l < nl; ++l)
    if (!line_status[l]) { topo_out[d->line_or_pos_topo_vect[l]] = -1; topo_out[d->line_ex_pos_topo_vect[l]] = -1; }
  for (int b = 0; b < nbt; ++b) nb += active[b];

  /* bus types */
  int nref = 0;
  for (int b = 0; b < nbt; ++b) { btype[b] = 0; vset[b] = 1.0; }
  for (int g = 0; g < ng; ++g) {
    if (gbus[g] < 0) continue;
    if (d->gen_slack[g]) btype[gbus[g]] = 2;
    else {
      if (btype[gbus[g]] != 2) btype[gbus[g]] = 1;
      P[gbus[g]] += inj[o.inj_gen_p + g] / sn;
    }
    vset[gbus[g]] = inj[o.inj_gen_vm + g];
  }
  for (int b = 0; b < nbt; ++b) nref += (active[b] && btype[b] == 2);
  if (nref == 0) { rc = GPF_ST_NOSLACK; goto done; }

  /* connectivity */
  {
    int* lab = (int*)calloc((size_t)nbt, sizeof(int));
    for (int b = 0; b < nbt; ++b) lab[b] = (active[b] && btype[b] == 2);
    int changed = 1;
    while (changed) {
      changed = 0;
      for (int l = 0; l < nl; ++l)
        if (line_status[l] && lab[lor[l]] != lab[lex[l]]) { lab[lor[l]] = lab[lex[l]] = 1; changed = 1; }
    }
    int bad = 0;
    for (int b = 0; b < nbt; ++b) bad |= (active[b] && !lab[b]);
    free(lab);
    if (bad) { rc = GPF_ST_ISLANDED; goto done; }
  }

  for (int i = 0; i < nd; ++i)
    if (lbus[i] >= 0) { Pd[lbus[i]] += inj[o.inj_load_p + i]; Qd[lbus[i]] += inj[o.inj_load_q + i]; }
  for (int i = 0; i < nst; ++i)
    if (sbus[i] >= 0) { Pd[sbus[i]] += inj[o.inj_sto_p + i]; Qd[sbus[i]] += inj[o.inj_sto_q + i]; }
  for (int i = 0; i < nsh; ++i)
    if (shb[i] >= 0) Gs[shb[i]] += inj[o.inj_sh_p + i] * d->shunt_fact[i] / sn;
  for (int b = 0; b < nbt; ++b) { P[b] -= Pd[b] / sn; Q[b] = -Qd[b] / sn; }

  int npvpq = 0, npq = 0;
  for (int b = 0; b < nbt; ++b) {
    pidx[b] = (active[b] && btype[b] != 2) ? npvpq++ : -1;
  }
  for (int b = 0; b < nbt; ++b) {
    qidx[b] = (active[b] && btype[b] == 0) ? npq++ : -1;
  }

  /* DC solve */
  {
    int n = npvpq;
    B = (double*)calloc((size_t)n * n + n + 1, sizeof(double));
    double* rhs = B + (size_t)n * n;
    for (int l = 0; l < nl; ++l) {
      if (!line_status[l]) continue;
      int f = lor[l], t = lex[l];
      double bb = d->br_bdc[l];
      int pf = pidx[f], pt = pidx[t];
      if (pf >= 0) B[pf * n + pf] += bb;
      if (pt >= 0) B[pt * n + pt] += bb;
      if (pf >= 0 && pt >= 0) { B[pf * n + pt] -= bb; B[pt * n + pf] -= bb; }
    }
    for (int b = 0; b < nbt; ++b)
      if (pidx[b] >= 0) rhs[pidx[b]] = P[b] - Gs[b];
    if (n > 0 && dense_solve(B, rhs, n)) { rc = GPF_ST_SINGULAR; goto done; }
    for (int b = 0; b < nbt; ++b) {
      va[b] = pidx[b] >= 0 ? rhs[pidx[b]] : 0.0;
      vm[b] = (btype[b] == 0) ? 1.0 : vset[b];
      if (!(fabs(va[b]) < 1e300)) { rc = GPF_ST_SINGULAR; goto done; }
    }
  }

  Y = (cplx*)calloc((size_t)nbt * nbt, sizeof(cplx));
  V = (cplx*)calloc((size_t)2 * nbt, sizeof(cplx));
  cplx* Ibus = V + nbt;
  if (!is_dc) {
    for (int l = 0; l < nl; ++l) {
      if (!line_status[l]) continue;
      int f = lor[l], t = lex[l];
      const double* y = d->br_y + 8 * (size_t)l;
      Y[f * nbt + f] += y[0] + I * y[1];
      Y[f * nbt + t] += y[2] + I * y[3];
      Y[t * nbt + f] += y[4] + I * y[5];
      Y[t * nbt + t] += y[6] + I * y[7];
    }
    for (int i = 0; i < nsh; ++i)
      if (shb[i] >= 0)
        Y[shb[i] * nbt + shb[i]] += (inj[o.inj_sh_p + i] - I * inj[o.inj_sh_q + i]) * d->shunt_fact[i] / sn;

    const int n = npvpq + npq;
    const double tol = tol_mva / sn;
    J = (double*)malloc(sizeof(double) * ((size_t)n * n + 1));
    F = (double*)malloc(sizeof(double) * ((size_t)n + 1));
    int converged = 0;
    for (;;) {
      for (int b = 0; b < nbt; ++b) V[b] = active[b] ? vm[b] * cexp(I * va[b]) : 0.0;
      double fmax = 0.0;
      int bad = 0;
      for (int b = 0; b < nbt; ++b) {
        if (!active[b]) continue;
        cplx ib = 0.0;
        for (int c = 0; c < nbt; ++c)
          if (active[c]) ib += Y[b * nbt + c] * V[c];
        Ibus[b] = ib;
        cplx s = V[b] * conj(ib);
        if (pidx[b] >= 0) {
          double mp = creal(s) - P[b];
          F[pidx[b]] = mp;
          if (!(fabs(mp) <= 1e300)) bad = 1;
          if (fabs(mp) > fmax) fmax = fabs(mp);
        }
        if (qidx[b] >= 0) {
          double mq = cimag(s) - Q[b];
          F[npvpq + qidx[b]] = mq;
          if (!(fabs(mq) <= 1e300)) bad = 1;
          if (fabs(mq) > fmax) fmax = fabs(mq);
        }
      }
      if (bad) { rc = GPF_ST_MAXITER; break; }
      if (fmax < tol) { converged = 1; break; }
      if (n_iter >= max_iter) break;
      ++n_iter;
      for (int i = 0; i < nbt; ++i) {
        if (pidx[i] < 0) continue;
        for (int j = 0; j < nbt; ++j) {
          if (pidx[j] < 0) continue;
          cplx T = V[i] * conj(Y[i * nbt + j] * V[j]);
          cplx dva, dvm;
          if (i == j) {
            cplx S = V[i] * conj(Ibus[i]);
            dva = I * (S - T);
            dvm = (T + S) / vm[j];
          } else {
            dva = -I * T;
            dvm = T / vm[j];
          }
          J[pidx[i] * n + pidx[j]] = creal(dva);
          if (qidx[j] >= 0) J[pidx[i] * n + npvpq + qidx[j]] = creal(dvm);
          if (qidx[i] >= 0) {
            J[(npvpq + qidx[i]) * n + pidx[j]] = cimag(dva);
            if (qidx[j] >= 0) J[(npvpq + qidx[i]) * n + npvpq + qidx[j]] = cimag(dvm);
          }
        }
      }
      for (int i = 0; i < n; ++i) F[i] = -F[i];
      if (dense_solve(J, F, n)) { rc = GPF_ST_SINGULAR; break; }
      for (int b = 0; b < nbt; ++b) {
        if (!active[b]) continue;
        if (pidx[b] >= 0) va[b] += F[pidx[b]];
        if (qidx[b] >= 0) vm[b] += F[npvpq + qidx[b]];
        if (vm[b] < 0.0) { vm[b] = -vm[b]; va[b] += M_PI; }
        va[b] = remainder(va[b], 2.0 * M_PI);
      }
    }
    if (rc == GPF_ST_CONVERGED && !converged) rc = GPF_ST_MAXITER;
    if (rc != GPF_ST_CONVERGED) goto done;
  }

  /* ---- results ------------------------------------------------------------------------------------ */
  {
    const double R2D = 57.295779513082320877, SQ3 = 1.7320508075688772935;
    double* Sre = (double*)calloc((size_t)2 * nbt, sizeof(double));
    double* Sim = Sre + nbt;
    if (is_dc) {
      for (int l = 0; l < nl; ++l) {
        if (!line_status[l]) continue;
        double fl = (va[lor[l]] - va[lex[l]]) * d->br_bdc[l];
        Sre[lor[l]] += fl;
        Sre[lex[l]] -= fl;
      }
      for (int b = 0; b < nbt; ++b) Sre[b] += Gs[b];
    } else {
      for (int b = 0; b < nbt; ++b) {
        if (!active[b]) continue;
        cplx s = V[b] * conj(Ibus[b]);
        Sre[b] = creal(s);
        Sim[b] = cimag(s);
      }
    }
    for (int i = 0; i < o.n_out; ++i) out[i] = 0.0;
    for (int l = 0; l < nl; ++l) {
      if (!line_status[l]) continue;
      int f = lor[l], t = lex[l];
      double vnf = d->sub_vn_kv[d->line_or_sub[l]], vnt = d->sub_vn_kv[d->line_ex_sub[l]];
      double pf, qf, pt, qt;
      if (is_dc) {
        pf = (va[f] - va[t]) * d->br_bdc[l] * sn; pt = -pf; qf = qt = 0.0;
      } else {
        const double* y = d->br_y + 8 * (size_t)l;
        cplx If = (y[0] + I * y[1]) * V[f] + (y[2] + I * y[3]) * V[t];
        cplx It = (y[4] + I * y[5]) * V[f] + (y[6] + I * y[7]) * V[t];
        cplx Sf = V[f] * conj(If) * sn, St = V[t] * conj(It) * sn;
        pf = creal(Sf); qf = cimag(Sf); pt = creal(St); qt = cimag(St);
      }
      out[o.p_or + l] = pf; out[o.q_or + l] = qf; out[o.p_ex + l] = pt; out[o.q_ex + l] = qt;
      out[o.a_or + l] = sqrt(pf * pf + qf * qf) / (SQ3 * vm[f] * vnf) * 1000.0;
      out[o.a_ex + l] = sqrt(pt * pt + qt * qt) / (SQ3 * vm[t] * vnt) * 1000.0;
      out[o.v_or + l] = vm[f] * vnf; out[o.v_ex + l] = vm[t] * vnt;
      out[o.th_or + l] = va[f] * R2D; out[o.th_ex + l] = va[t] * R2D;
    }
    for (int i = 0; i < nd; ++i) {
      if (lbus[i] < 0) continue;
      out[o.load_p + i] = inj[o.inj_load_p + i];
      out[o.load_q + i] = is_dc ? 0.0 : inj[o.inj_load_q + i];
      out[o.load_v + i] = vm[lbus[i]] * d->sub_vn_kv[d->load_sub[i]];
      out[o.load_th + i] = va[lbus[i]] * R2D;
    }
    for (int i = 0; i < nst; ++i) {
      if (sbus[i] < 0) continue;
      out[o.sto_p + i] = inj[o.inj_sto_p + i];
      out[o.sto_q + i] = is_dc ? 0.0 : inj[o.inj_sto_q + i];
      out[o.sto_v + i] = vm[sbus[i]] * d->sub_vn_kv[d->storage_sub[i]];
      out[o.sto_th + i] = va[sbus[i]] * R2D;
    }
    for (int i = 0; i < nsh; ++i) {
      shunt_bus_out[i] = shb[i] >= 0 ? shunt_bus[i] : -1;
      if (shb[i] < 0) continue;
      double v = vm[shb[i]];
      out[o.sh_p + i] = inj[o.inj_sh_p + i] * d->shunt_fact[i] * v * v;
      out[o.sh_q + i] = is_dc ? 0.0 : inj[o.inj_sh_q + i] * d->shunt_fact[i] * v * v;
      out[o.sh_v + i] = v * d->sub_vn_kv[d->shunt_sub[i]];
    }
    for (int g = 0; g < ng; ++g) {
      int b = gbus[g];
      if (b < 0) continue;
      int cnt = 0, nslack = 0;
      double qmin_t = 0, qmax_t = 0, p_others = 0;
      for (int k = 0; k < ng; ++k)
        if (gbus[k] == b) {
          ++cnt; qmin_t += d->gen_min_q[k]; qmax_t += d->gen_max_q[k];
          if (d->gen_slack[k]) ++nslack; else p_others += inj[o.inj_gen_p + k];
        }
      double qtot = Sim[b] * sn + Qd[b], q;
      if (is_dc) q = 0.0;
      else if (cnt == 1) q = qtot;
      else if (qmin_t == qmax_t) q = qtot / cnt;
      else q = d->gen_min_q[g] + (qtot - qmin_t) / (qmax_t - qmin_t + 2.220446049250313e-16) * (d->gen_max_q[g] - d->gen_min_q[g]);
      double p = inj[o.inj_gen_p + g];
      if (d->gen_slack[g]) p = (Sre[b] * sn + Pd[b] - p_others) / nslack;
      out[o.gen_p + g] = p; out[o.gen_q + g] = q;
      out[o.gen_v + g] = vm[b] * d->sub_vn_kv[d->gen_sub[g]];
      out[o.gen_th + g] = va[b] * R2D;
    }
    for (int b = 0; b < nbt; ++b) {
      bus_vm[b] = active[b] ? vm[b] : NAN;
      bus_va[b] = active[b] ? va[b] * R2D : NAN;
    }
    free(Sre);
  }

done:
  if (rc != GPF_ST_CONVERGED) fill_fail(d, &o, out, topo_out, shunt_bus_out, line_status, bus_vm, bus_va);
  status4[0] = rc; status4[1] = n_iter; status4[2] = nb; status4[3] = 0;
  free(lor); free(P); free(Y); free(V); free(J); free(F); free(B);
  return rc;
}

/* The synthetic DoNothing step of bench.py for lanes [lane0, lane0+n): chronics row -> injections
 * (float32 API values) -> AC power flow.  Returns the number of converged lanes. `out` may be NULL
 * (timing only).  Mirrors gpf::step_kernel's K9 stage. */
int pfo_step_batch(const gpf_grid_desc* d, const float* chron, int T, const int32_t* lane_offset, const float* lane_scale,
                   double rebalance, int t, int lane0, int n, int max_iter, double tol_mva, double* out_rows,
                   int32_t* status_rows) {
  const offsets_t o = make_offsets(d);
  const int nd = d->n_load, ng = d->n_gen, nch = 2 * nd + 2 * ng, nbt = d->n_sub * d->n_busbar;
  double* inj = (double*)malloc(sizeof(double) * (size_t)o.n_inj);
  double* out = (double*)malloc(sizeof(double) * (size_t)o.n_out);
  int32_t* topo_out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(d->dim_topo + d->n_shunt + 4));
  int32_t* sb_out = topo_out + d->dim_topo;
  int32_t st[4];
  uint8_t* ls = (uint8_t*)malloc((size_t)d->n_line + 1);
  double* bvm = (double*)malloc(sizeof(double) * (size_t)2 * nbt);
  int nconv = 0;
  for (int k = lane0; k < lane0 + n; ++k) {
    memcpy(inj, d->init_inj, sizeof(double) * (size_t)o.n_inj);
    int row = (t + (lane_offset ? lane_offset[k] : 0)) % T;
    if (row < 0) row += T;
    const float* ch = chron + (size_t)row * nch;
    const float* sc = lane_scale ? lane_scale + (size_t)k * 2 * nd : NULL;
    double sl = 0, sp = 0;
    for (int i = 0; i < nd; ++i) {
      float lp = ch[i], lq = ch[nd + i];
      if (sc) { lp *= sc[i]; lq *= sc[nd + i]; }
      inj[o.inj_load_p + i] = lp; inj[o.inj_load_q + i] = lq; sl += lp;
    }
    for (int i = 0; i < ng; ++i) if (!d->gen_slack[i]) sp += ch[2 * nd + i];
    float fac = (rebalance > 0 && sp > 0) ? (float)(rebalance * sl / sp) : 1.0f;
    for (int i = 0; i < ng; ++i) {
      float pp = ch[2 * nd + i];
      if (!d->gen_slack[i]) pp *= fac;
      float vn = (float)d->sub_vn_kv[d->gen_sub[i]];
      inj[o.inj_gen_p + i] = pp;
      inj[o.inj_gen_vm + i] = (double)(ch[2 * nd + ng + i] / vn);
    }
    double* orow = out_rows ? out_rows + (size_t)(k - lane0) * o.n_out : out;
    pfo_solve(d, inj, d->init_topo, d->init_shunt_bus, 0, max_iter, tol_mva, orow, topo_out, sb_out, ls, st, bvm, bvm + nbt);
    if (status_rows) memcpy(status_rows + (size_t)(k - lane0) * 4, st, sizeof(st));
    nconv += (st[0] == GPF_ST_CONVERGED);
  }
  free(inj); free(out); free(topo_out); free(ls); free(bvm);
  return nconv;
}
