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

/* ---- SPARSE variant of the two linear solves (pfo_set_solver(1)) ---------------------------------------------------------------
 * What a sparse CPU power flow does -- pandapower's newtonpf calls scipy.sparse.linalg.spsolve per iteration
 * (grid2op/Backend/pandaPowerBackend.py:1081-1083 picks the solver), lightsim2grid factorises with KLU and keeps the symbolic
 * analysis --: LU without pivoting on a minimum-degree ordering of the bus graph (power-flow Jacobians are factorised that way by
 * every production solver; a tiny pivot falls back to the dense elimination above), fill pattern computed once per
 * (topology, bus types) and reused by every iteration and every solve with the same pattern.  Same Newton iteration as the dense
 * path: tests/test_oracle_c.py pins the two against each other (1e-10) on every golden grid and on random topologies.  Used by
 * bench.py's per-config cpu_baseline -- the dense path is a straw man on 118 substations (1.4 ms per power flow). */
typedef struct {
  int n, nnz, cap_n, cap_nnz;
  int *rp, *ci, *dpos;      /* CSR of L + U in the permuted numbering, columns sorted, diagonal position per row */
  double *val, *w;          /* values, dense work row */
  int* mark;
} splu_t;

static void __attribute__((unused)) splu_free(splu_t* s) { free(s->rp); free(s->ci); free(s->dpos); free(s->val); free(s->w); free(s->mark); memset(s, 0, sizeof(*s)); }

/* pattern of A given as rows of (permuted) column indices arp / aci (unsorted, duplicates allowed) -> filled pattern of L + U */
static int splu_symbolic(splu_t* s, int n, const int* arp, const int* aci) {
  if (n > s->cap_n) {
    free(s->rp); free(s->dpos); free(s->w); free(s->mark);
    s->cap_n = n + 16;
    s->rp = (int*)malloc(sizeof(int) * (size_t)(s->cap_n + 1));
    s->dpos = (int*)malloc(sizeof(int) * (size_t)s->cap_n);
    s->w = (double*)malloc(sizeof(double) * (size_t)s->cap_n);
    s->mark = (int*)malloc(sizeof(int) * (size_t)s->cap_n);
    if (!s->rp || !s->dpos || !s->w || !s->mark) return 1;
  }
  s->n = n;
  int nnz = 0;
  s->rp[0] = 0;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) s->mark[j] = 0;
    s->mark[i] = 1;
    for (int q = arp[i]; q < arp[i + 1]; ++q) s->mark[aci[q]] = 1;
    /* row i takes the pattern right of the diagonal of every row k < i it has an entry in (ascending k: fill adds more of them) */
    for (int k = 0; k < i; ++k) {
      if (!s->mark[k]) continue;
      for (int q = s->dpos[k] + 1; q < s->rp[k + 1]; ++q) s->mark[s->ci[q]] = 1;
    }
    int cnt = 0;
    for (int j = 0; j < n; ++j) cnt += s->mark[j];
    if (nnz + cnt > s->cap_nnz) {
      s->cap_nnz = 2 * (nnz + cnt) + 1024;
      s->ci = (int*)realloc(s->ci, sizeof(int) * (size_t)s->cap_nnz);
      s->val = (double*)realloc(s->val, sizeof(double) * (size_t)s->cap_nnz);
      if (!s->ci || !s->val) return 1;
    }
    for (int j = 0; j < n; ++j)
      if (s->mark[j]) { if (j == i) s->dpos[i] = nnz; s->ci[nnz++] = j; }
    s->rp[i + 1] = nnz;
  }
  s->nnz = nnz;
  return 0;
}
static inline int splu_pos(const splu_t* s, int r, int c) {      /* position of (r, c): binary search in the sorted row */
  int lo = s->rp[r], hi = s->rp[r + 1] - 1;
  while (lo <= hi) { int mid = (lo + hi) >> 1; if (s->ci[mid] == c) return mid; if (s->ci[mid] < c) lo = mid + 1; else hi = mid - 1; }
  return -1;
}
/* in-place LU of the values in s->val (row-wise IKJ elimination over the filled pattern); 1: a pivot is unusable */
static int splu_factor(splu_t* s) {
  const int n = s->n;
  double* w = s->w;
  for (int i = 0; i < n; ++i) {
    double rmax = 0.0;
    for (int q = s->rp[i]; q < s->rp[i + 1]; ++q) { w[s->ci[q]] = s->val[q]; if (fabs(s->val[q]) > rmax) rmax = fabs(s->val[q]); }
    for (int q = s->rp[i]; q < s->dpos[i]; ++q) {
      const int k = s->ci[q];
      const double l = w[k] / s->val[s->dpos[k]];
      w[k] = l;
      if (l != 0.0) for (int r = s->dpos[k] + 1; r < s->rp[k + 1]; ++r) w[s->ci[r]] -= l * s->val[r];
    }
    for (int q = s->rp[i]; q < s->rp[i + 1]; ++q) s->val[q] = w[s->ci[q]];
    const double piv = s->val[s->dpos[i]];
    if (!(fabs(piv) > 1e-11 * rmax) || !(fabs(piv) > 1e-300) || !(fabs(piv) < 1e300)) return 1;
  }
  return 0;
}
static void splu_solve(const splu_t* s, double* b) {              /* b (permuted numbering) -> solution */
  const int n = s->n;
  for (int i = 0; i < n; ++i) { double t = b[i]; for (int q = s->rp[i]; q < s->dpos[i]; ++q) t -= s->val[q] * b[s->ci[q]]; b[i] = t; }
  for (int i = n - 1; i >= 0; --i) {
    double t = b[i];
    for (int q = s->dpos[i] + 1; q < s->rp[i + 1]; ++q) t -= s->val[q] * b[s->ci[q]];
    b[i] = t / s->val[s->dpos[i]];
  }
}

/* per-thread solver state: the bus graph, the elimination order and the two symbolic factorisations of the last pattern seen */
typedef struct {
  int sparse;
  uint64_t key; int have;
  int nbt, *ap, *ai;            /* bus adjacency (CSR incl. the diagonal) over the in-service lines */
  int *order, *upos, *rank;     /* elimination order of the buses with unknowns; first permuted AC unknown (theta) of each bus and its DC unknown, -1: none */
  int cap_b, cap_a;
  splu_t dc, ac;
  cplx* yv; int cap_y;          /* Ybus values aligned with ai */
  int *pat_rp, *pat_ci; int cap_pr, cap_pc;
} spctx_t;
static __thread spctx_t g_sp;
void pfo_set_solver(int sparse) { g_sp.sparse = sparse != 0; }
int pfo_get_solver(void) { return g_sp.sparse; }

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
#define PFO_RETRY_DENSE (-77)      /* sparse path: a pivot was unusable -> the caller repeats the solve with the dense elimination */
static uint64_t sp_hash(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}
/* (re)build the bus graph, the elimination order and both symbolic factorisations for the pattern at hand; 0 on success */
static int sp_prepare(spctx_t* c, int nbt, int nl, const int* lor, const int* lex, const uint8_t* line_status, const int* active,
                      const int* pidx, const int* qidx, int npvpq, int npq) {
  uint64_t key = 1469598103934665603ull;
  key = sp_hash(key, &nbt, sizeof(nbt)); key = sp_hash(key, lor, sizeof(int) * (size_t)nl); key = sp_hash(key, lex, sizeof(int) * (size_t)nl);
  key = sp_hash(key, line_status, (size_t)nl); key = sp_hash(key, pidx, sizeof(int) * (size_t)nbt); key = sp_hash(key, qidx, sizeof(int) * (size_t)nbt);
  key = sp_hash(key, active, sizeof(int) * (size_t)nbt);
  if (c->have && c->key == key && c->nbt == nbt) return 0;
  c->have = 0;
  if (nbt > c->cap_b) {
    free(c->ap); free(c->order); free(c->upos); free(c->rank);
    c->cap_b = nbt + 16;
    c->ap = (int*)malloc(sizeof(int) * (size_t)(c->cap_b + 1));
    c->order = (int*)malloc(sizeof(int) * (size_t)c->cap_b);
    c->upos = (int*)malloc(sizeof(int) * (size_t)c->cap_b);
    c->rank = (int*)malloc(sizeof(int) * (size_t)c->cap_b);
  }
  if (2 * nl + nbt > c->cap_a) { free(c->ai); c->cap_a = 2 * nl + nbt + 64; c->ai = (int*)malloc(sizeof(int) * (size_t)c->cap_a); }
  if (!c->ap || !c->order || !c->upos || !c->rank || !c->ai) return 1;
  c->nbt = nbt;
  unsigned char* adj = (unsigned char*)calloc((size_t)nbt * nbt, 1);
  int* deg = (int*)calloc((size_t)2 * nbt, sizeof(int));
  int* gone = deg + nbt;
  if (!adj || !deg) { free(adj); free(deg); return 1; }
  for (int l = 0; l < nl; ++l)
    if (line_status[l] && lor[l] != lex[l]) { adj[lor[l] * nbt + lex[l]] = 1; adj[lex[l] * nbt + lor[l]] = 1; }
  /* adjacency over ALL active buses (Ibus needs the reference buses too), diagonal first */
  int na = 0;
  for (int b = 0; b < nbt; ++b) {
    c->ap[b] = na;
    if (!active[b]) continue;
    c->ai[na++] = b;
    for (int j = 0; j < nbt; ++j) if (adj[b * nbt + j]) c->ai[na++] = j;
  }
  c->ap[nbt] = na;
  /* minimum-degree order of the buses that carry unknowns (reference buses drop out of the reduced systems) */
  for (int b = 0; b < nbt; ++b) {
    gone[b] = pidx[b] < 0;
    if (gone[b]) for (int j = 0; j < nbt; ++j) { adj[b * nbt + j] = 0; adj[j * nbt + b] = 0; }
  }
  for (int b = 0; b < nbt; ++b) { deg[b] = 0; for (int j = 0; j < nbt; ++j) deg[b] += adj[b * nbt + j]; }
  int* nbv = (int*)malloc(sizeof(int) * (size_t)nbt);
  if (!nbv) { free(adj); free(deg); return 1; }
  for (int k = 0; k < npvpq; ++k) {
    int best = -1;
    for (int b = 0; b < nbt; ++b) if (!gone[b] && (best < 0 || deg[b] < deg[best])) best = b;
    c->order[k] = best;
    gone[best] = 1;
    int nn = 0;
    for (int j = 0; j < nbt; ++j) if (adj[best * nbt + j]) { nbv[nn++] = j; adj[best * nbt + j] = 0; adj[j * nbt + best] = 0; --deg[j]; }
    for (int x = 0; x < nn; ++x)
      for (int y = x + 1; y < nn; ++y)
        if (!adj[nbv[x] * nbt + nbv[y]]) { adj[nbv[x] * nbt + nbv[y]] = 1; adj[nbv[y] * nbt + nbv[x]] = 1; ++deg[nbv[x]]; ++deg[nbv[y]]; }
  }
  free(nbv); free(adj); free(deg);
  /* permuted numbering: DC unknown of bus b = its rank in the order; AC: theta_b at upos[b], |V|_b right behind it (PQ buses) */
  int* const rank = c->rank;
  for (int b = 0; b < nbt; ++b) { rank[b] = -1; c->upos[b] = -1; }
  int u = 0;
  for (int k = 0; k < npvpq; ++k) { const int b = c->order[k]; rank[b] = k; c->upos[b] = u; u += 1 + (qidx[b] >= 0); }
  /* patterns of the two systems */
  const int n_ac = npvpq + npq;
  const size_t need_c = (size_t)4 * (size_t)na + 16;
  if (n_ac + 1 > c->cap_pr) { free(c->pat_rp); c->cap_pr = n_ac + 17; c->pat_rp = (int*)malloc(sizeof(int) * (size_t)c->cap_pr); }
  if ((int)need_c > c->cap_pc) { free(c->pat_ci); c->cap_pc = (int)need_c; c->pat_ci = (int*)malloc(sizeof(int) * need_c); }
  if (!c->pat_rp || !c->pat_ci) return 1;
  int q = 0;
  for (int k = 0; k < npvpq; ++k) {                      /* DC: one row per bus */
    const int b = c->order[k];
    c->pat_rp[k] = q;
    for (int a = c->ap[b]; a < c->ap[b + 1]; ++a) if (rank[c->ai[a]] >= 0) c->pat_ci[q++] = rank[c->ai[a]];
  }
  c->pat_rp[npvpq] = q;
  int rc = splu_symbolic(&c->dc, npvpq, c->pat_rp, c->pat_ci);
  q = 0;
  int r = 0;
  for (int k = 0; k < npvpq && !rc; ++k) {               /* AC: the theta row and (PQ buses) the |V| row of a bus share their columns */
    const int b = c->order[k];
    for (int rep = 0; rep < 1 + (qidx[b] >= 0); ++rep) {
      c->pat_rp[r++] = q;
      for (int a = c->ap[b]; a < c->ap[b + 1]; ++a) {
        const int j = c->ai[a];
        if (c->upos[j] < 0) continue;
        c->pat_ci[q++] = c->upos[j];
        if (qidx[j] >= 0) c->pat_ci[q++] = c->upos[j] + 1;
      }
    }
  }
  c->pat_rp[r] = q;
  if (!rc) rc = splu_symbolic(&c->ac, n_ac, c->pat_rp, c->pat_ci);
  if (na > c->cap_y) { free(c->yv); c->cap_y = na + 64; c->yv = (cplx*)malloc(sizeof(cplx) * (size_t)c->cap_y); if (!c->yv) rc = 1; }
  if (rc) return 1;
  c->key = key; c->have = 1;
  return 0;
}
static inline int sp_apos(const spctx_t* c, int b, int j) {          /* position of (b, j) in the bus adjacency */
  for (int a = c->ap[b]; a < c->ap[b + 1]; ++a) if (c->ai[a] == j) return a;
  return -1;
}

static int pfo_solve_impl(const gpf_grid_desc* d, const double* inj, const int32_t* topo, const int32_t* shunt_bus, int is_dc,
              int max_iter, double tol_mva, double* out, int32_t* topo_out, int32_t* shunt_bus_out, uint8_t* line_status,
              int32_t* status4, double* bus_vm, double* bus_va, const int sparse) {
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

  spctx_t* const sp = &g_sp;
  if (sparse && sp_prepare(sp, nbt, nl, lor, lex, line_status, active, pidx, qidx, npvpq, npq)) { rc = PFO_RETRY_DENSE; goto done; }
  /* DC solve */
  if (sparse) {
    splu_t* L = &sp->dc;
    double* rhs = (double*)malloc(sizeof(double) * (size_t)(npvpq + 1));
    for (int q = 0; q < L->nnz; ++q) L->val[q] = 0.0;
    for (int l = 0; l < nl; ++l) {
      if (!line_status[l]) continue;
      const int f = lor[l], t = lex[l];
      const double bb = d->br_bdc[l];
      const int kf = sp->rank[f], kt = sp->rank[t];                                 /* DC unknowns of the two ends, -1: a reference bus */
      if (kf >= 0) L->val[L->dpos[kf]] += bb;
      if (kt >= 0) L->val[L->dpos[kt]] += bb;
      if (kf >= 0 && kt >= 0 && f != t) { L->val[splu_pos(L, kf, kt)] -= bb; L->val[splu_pos(L, kt, kf)] -= bb; }
      else if (kf >= 0 && kt >= 0) { L->val[L->dpos[kf]] -= 2.0 * bb; }            /* (a line between the two busbars' same bus: cancels) */
    }
    for (int k = 0; k < npvpq; ++k) { const int b = sp->order[k]; rhs[k] = P[b] - Gs[b]; }
    if (npvpq > 0) {
      if (splu_factor(L)) { free(rhs); rc = PFO_RETRY_DENSE; goto done; }
      splu_solve(L, rhs);
    }
    for (int b = 0; b < nbt; ++b) { va[b] = 0.0; vm[b] = (btype[b] == 0) ? 1.0 : vset[b]; }
    for (int k = 0; k < npvpq; ++k) {
      va[sp->order[k]] = rhs[k];
      if (!(fabs(rhs[k]) < 1e300)) { free(rhs); rc = GPF_ST_SINGULAR; goto done; }
    }
    free(rhs);
  } else {
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

  V = (cplx*)calloc((size_t)2 * nbt, sizeof(cplx));
  cplx* Ibus = V + nbt;
  if (!is_dc && sparse) {
    /* Ybus over the bus adjacency */
    cplx* const yv = sp->yv;
    const int na = sp->ap[nbt];
    for (int a = 0; a < na; ++a) yv[a] = 0.0;
    for (int l = 0; l < nl; ++l) {
      if (!line_status[l]) continue;
      const int f = lor[l], t = lex[l];
      const double* y = d->br_y + 8 * (size_t)l;
      yv[sp->ap[f]] += y[0] + I * y[1];
      yv[sp_apos(sp, f, t)] += y[2] + I * y[3];
      yv[sp_apos(sp, t, f)] += y[4] + I * y[5];
      yv[sp->ap[t]] += y[6] + I * y[7];
    }
    for (int i = 0; i < nsh; ++i)
      if (shb[i] >= 0) yv[sp->ap[shb[i]]] += (inj[o.inj_sh_p + i] - I * inj[o.inj_sh_q + i]) * d->shunt_fact[i] / sn;
    splu_t* L = &sp->ac;
    const int n = npvpq + npq;
    const double tol = tol_mva / sn;
    F = (double*)malloc(sizeof(double) * ((size_t)n + 1));
    int converged = 0;
    for (;;) {
      for (int b = 0; b < nbt; ++b) V[b] = active[b] ? vm[b] * cexp(I * va[b]) : 0.0;
      double fmax = 0.0;
      int bad = 0;
      for (int b = 0; b < nbt; ++b) {
        if (!active[b]) continue;
        cplx ib = 0.0;
        for (int a = sp->ap[b]; a < sp->ap[b + 1]; ++a) ib += yv[a] * V[sp->ai[a]];
        Ibus[b] = ib;
        const cplx s_ = V[b] * conj(ib);
        const int u = sp->upos[b];
        if (u >= 0) {
          const double mp = creal(s_) - P[b];
          F[u] = mp;
          if (!(fabs(mp) <= 1e300)) bad = 1;
          if (fabs(mp) > fmax) fmax = fabs(mp);
          if (qidx[b] >= 0) {
            const double mq = cimag(s_) - Q[b];
            F[u + 1] = mq;
            if (!(fabs(mq) <= 1e300)) bad = 1;
            if (fabs(mq) > fmax) fmax = fabs(mq);
          }
        }
      }
      if (bad) { rc = GPF_ST_MAXITER; break; }
      if (fmax < tol) { converged = 1; break; }
      if (n_iter >= max_iter) break;
      ++n_iter;
      for (int q = 0; q < L->nnz; ++q) L->val[q] = 0.0;
      for (int i = 0; i < nbt; ++i) {
        const int ri = sp->upos[i];
        if (ri < 0) continue;
        const int iq = qidx[i] >= 0;
        for (int a = sp->ap[i]; a < sp->ap[i + 1]; ++a) {
          const int j = sp->ai[a], cj = sp->upos[j];
          if (cj < 0) continue;
          const cplx T = V[i] * conj(yv[a] * V[j]);
          cplx dva, dvm;
          if (i == j) {
            const cplx S = V[i] * conj(Ibus[i]);
            dva = I * (S - T);
            dvm = (T + S) / vm[j];
          } else {
            dva = -I * T;
            dvm = T / vm[j];
          }
          const int p0 = splu_pos(L, ri, cj);
          L->val[p0] = creal(dva);
          if (qidx[j] >= 0) L->val[p0 + 1] = creal(dvm);              /* (columns sorted: the |V| column of a bus follows its theta column) */
          if (iq) {
            const int p1 = splu_pos(L, ri + 1, cj);
            L->val[p1] = cimag(dva);
            if (qidx[j] >= 0) L->val[p1 + 1] = cimag(dvm);
          }
        }
      }
      for (int i = 0; i < n; ++i) F[i] = -F[i];
      if (splu_factor(L)) { rc = PFO_RETRY_DENSE; break; }
      splu_solve(L, F);
      for (int b = 0; b < nbt; ++b) {
        if (!active[b]) continue;
        const int u = sp->upos[b];
        if (u >= 0) { va[b] += F[u]; if (qidx[b] >= 0) vm[b] += F[u + 1]; }
        if (vm[b] < 0.0) { vm[b] = -vm[b]; va[b] += M_PI; }
        va[b] = remainder(va[b], 2.0 * M_PI);
      }
    }
    if (rc == GPF_ST_CONVERGED && !converged) rc = GPF_ST_MAXITER;
    if (rc != GPF_ST_CONVERGED) goto done;
  } else if (!is_dc) {
    Y = (cplx*)calloc((size_t)nbt * nbt, sizeof(cplx));
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
  if (rc == PFO_RETRY_DENSE) { free(lor); free(P); free(Y); free(V); free(J); free(F); free(B); return rc; }
  if (rc != GPF_ST_CONVERGED) fill_fail(d, &o, out, topo_out, shunt_bus_out, line_status, bus_vm, bus_va);
  status4[0] = rc; status4[1] = n_iter; status4[2] = nb; status4[3] = 0;
  free(lor); free(P); free(Y); free(V); free(J); free(F); free(B);
  return rc;
}

int pfo_solve(const gpf_grid_desc* d, const double* inj, const int32_t* topo, const int32_t* shunt_bus, int is_dc,
              int max_iter, double tol_mva, double* out, int32_t* topo_out, int32_t* shunt_bus_out, uint8_t* line_status,
              int32_t* status4, double* bus_vm, double* bus_va) {
  if (g_sp.sparse) {
    const int rc = pfo_solve_impl(d, inj, topo, shunt_bus, is_dc, max_iter, tol_mva, out, topo_out, shunt_bus_out, line_status, status4, bus_vm,
                                  bus_va, 1);
    if (rc != PFO_RETRY_DENSE) return rc;
  }
  return pfo_solve_impl(d, inj, topo, shunt_bus, is_dc, max_iter, tol_mva, out, topo_out, shunt_bus_out, line_status, status4, bus_vm, bus_va, 0);
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
