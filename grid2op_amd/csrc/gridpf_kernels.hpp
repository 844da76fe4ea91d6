// gridpf_kernels.hpp -- device code of the batched AC/DC power-flow engine (gfx950 / CDNA4 only).
//
// Mapping: ONE WAVEFRONT (64 lanes of the SIMD) PER GRID INSTANCE ("lane" in the C ABI; called
// "instance" here to avoid the clash with SIMD lanes).  A workgroup is exactly one wavefront, so
// __syncthreads() is a single-wave barrier.  All per-instance state (bus maps, voltages, injections,
// dense Ybus, dense Jacobian) is staged in LDS; the static grid tables are read through the scalar /
// vector caches (they are identical for every instance and stay L2 resident); the per-instance input
// and output rows are instance-major so that the 64 lanes of the wave touch consecutive addresses.
//
// Pipeline per instance (reference counterparts, paths relative to the reference checkout):
//   K1 topology compaction      PandaPowerBackend.apply_action bus scatter + pandapower pd2ppc bus lookup
//                               (grid2op/Backend/pandaPowerBackend.py:920-975)
//   K2 Ybus assembly            pandapower makeYbus (SURVEY.md A4')
//   K3 DC solve                 runpp(init="dc") / rundcpp  (pandaPowerBackend.py:1086-1090)
//   K4 mismatch + Jacobian      pypower newtonpf / dSbus_dV
//   K5 dense LU + solve         scipy.sparse.linalg.spsolve per Newton iteration (:1081-1083)
//   K6 result extraction        _fetch_data_pf_converged + pypower pfsoln (:1122-1218)
//   K7 overflow / cascade       Backend.next_grid_state (grid2op/Backend/backend.py:1476-1520)
//   K9 chronics gather          chronics_handler.next_time_step (grid2op/Environment/baseEnv.py:2516-2563)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpf {

constexpr int WAVE = 64;

struct GridDev {
  int n_sub, n_busbar, nb_tot, n_line, n_gen, n_load, n_sto, n_shunt, dim_topo;
  int n_inj, n_out, n_chron;
  double sn_mva;
  const double* sub_vn_kv;
  const int* line_or_sub;
  const int* line_ex_sub;
  const int* line_or_pos;
  const int* line_ex_pos;
  const double* br_y;      // [n_line][8]
  const double* br_bdc;    // [n_line]
  const int* gen_sub;
  const int* gen_pos;
  const double* gen_min_q;
  const double* gen_max_q;
  const unsigned char* gen_slack;
  const int* load_sub;
  const int* load_pos;
  const int* sto_sub;
  const int* sto_pos;
  const int* shunt_sub;
  const double* shunt_fact;
};

struct Bufs {
  double* inj;                 // [B][n_inj]
  int* topo;                   // [B][dim_topo]
  int* shunt_bus;              // [B][n_shunt]
  float* out;                  // [B][n_out]
  int* topo_out;               // [B][dim_topo]
  int* shunt_bus_out;          // [B][n_shunt]
  unsigned char* line_status;  // [B][n_line]
  int* status;                 // [B][4]
  double* bus_vm;              // [B][nb_tot]
  double* bus_va;              // [B][nb_tot]
  double* work;                // [B][work_stride]  (only used by the BIG variant)
  long long work_stride;
  // stepping
  const float* chron;          // [n_tab][T][n_chron]
  const int* lane_table;       // [B]
  const int* lane_offset;      // [B]
  const float* lane_scale;     // [B][2*n_load] or nullptr
  const float* thermal_limit;  // [n_line]
  float* rho;                  // [B][n_line]
  int* overflow_count;         // [B][n_line]
  int* disc_round;             // [B][n_line]
};

struct StepArgs {
  int t, T, rebalance_on, cascade, nb_ts_allowed, max_rounds, is_dc;
  double rebalance;
  float hard_overflow, soft_overflow;
};

// results-row offsets (must match gpf_layout in include/gridpf.h)
struct OutOff {
  int p_or, q_or, v_or, a_or, th_or, p_ex, q_ex, v_ex, a_ex, th_ex;
  int gen_p, gen_q, gen_v, gen_th, load_p, load_q, load_v, load_th, sto_p, sto_q, sto_v, sto_th, sh_p, sh_q, sh_v;
  int inj_gen_p, inj_gen_vm, inj_load_p, inj_load_q, inj_sto_p, inj_sto_q, inj_sh_p, inj_sh_q;
};

__host__ __device__ inline OutOff make_offsets(int nl, int ng, int nd, int ns, int nsh) {
  OutOff o;
  int k = 0;
  o.p_or = k; k += nl; o.q_or = k; k += nl; o.v_or = k; k += nl; o.a_or = k; k += nl; o.th_or = k; k += nl;
  o.p_ex = k; k += nl; o.q_ex = k; k += nl; o.v_ex = k; k += nl; o.a_ex = k; k += nl; o.th_ex = k; k += nl;
  o.gen_p = k; k += ng; o.gen_q = k; k += ng; o.gen_v = k; k += ng; o.gen_th = k; k += ng;
  o.load_p = k; k += nd; o.load_q = k; k += nd; o.load_v = k; k += nd; o.load_th = k; k += nd;
  o.sto_p = k; k += ns; o.sto_q = k; k += ns; o.sto_v = k; k += ns; o.sto_th = k; k += ns;
  o.sh_p = k; k += nsh; o.sh_q = k; k += nsh; o.sh_v = k; k += nsh;
  k = 0;
  o.inj_gen_p = k; k += ng; o.inj_gen_vm = k; k += ng; o.inj_load_p = k; k += nd; o.inj_load_q = k; k += nd;
  o.inj_sto_p = k; k += ns; o.inj_sto_q = k; k += ns; o.inj_sh_p = k; k += nsh; o.inj_sh_q = k; k += nsh;
  return o;
}

// bus types
constexpr int BT_PQ = 0, BT_PV = 1, BT_REF = 2;

// ---------------------------------------------------------------------------------------------------
// LDS carve.  Doubles first (8-byte aligned), then ints.  Host mirror: lds_bytes().
struct Carve {
  // doubles [nbc]
  double *vm, *va, *e, *f, *Psp, *Qsp, *Sre, *Sim, *vset, *Pd, *Qd, *Gs;
  double *Y;   // [nbc*nbc*2]   (LDS or global)
  double *J;   // [nJ*(nJ+1)]   (LDS or global)
  // ints
  int *gmap;    // [nb_tot] global bus -> compact index (-1 inactive)
  int *gid;     // [nbc] compact -> global
  int *btype, *pidx, *qidx, *lab;   // [nbc]
  int *lor_c, *lex_c;               // [n_line]
  int *gen_c, *load_c, *sto_c, *sh_c;
  int *misc;                        // [8] scratch scalars
};

__host__ __device__ inline size_t lds_bytes(const GridDev& g, int nbc, int nJ, bool big) {
  size_t nd = (size_t)12 * nbc;
  if (!big) nd += (size_t)2 * nbc * nbc + (size_t)nJ * (nJ + 1);
  size_t ni = (size_t)g.nb_tot + 5 * (size_t)nbc + 2 * (size_t)g.n_line + g.n_gen + g.n_load + g.n_sto + g.n_shunt + 8;
  return nd * 8 + ni * 4 + 16;
}

__device__ inline void carve_lds(Carve& c, unsigned char* base, const GridDev& g, int nbc, int nJ, double* work_big) {
  double* d = reinterpret_cast<double*>(base);
  c.vm = d; d += nbc; c.va = d; d += nbc; c.e = d; d += nbc; c.f = d; d += nbc;
  c.Psp = d; d += nbc; c.Qsp = d; d += nbc; c.Sre = d; d += nbc; c.Sim = d; d += nbc;
  c.vset = d; d += nbc; c.Pd = d; d += nbc; c.Qd = d; d += nbc; c.Gs = d; d += nbc;
  if (work_big) {
    c.Y = work_big;
    c.J = work_big + (size_t)2 * nbc * nbc;
  } else {
    c.Y = d; d += (size_t)2 * nbc * nbc;
    c.J = d; d += (size_t)nJ * (nJ + 1);
  }
  int* i = reinterpret_cast<int*>(d);
  c.gmap = i; i += g.nb_tot;
  c.gid = i; i += nbc; c.btype = i; i += nbc; c.pidx = i; i += nbc; c.qidx = i; i += nbc; c.lab = i; i += nbc;
  c.lor_c = i; i += g.n_line; c.lex_c = i; i += g.n_line;
  c.gen_c = i; i += g.n_gen; c.load_c = i; i += g.n_load; c.sto_c = i; i += g.n_sto; c.sh_c = i; i += g.n_shunt;
  c.misc = i;
}

// ---------------------------------------------------------------------------------------------------
__device__ inline double wave_max(double v) {
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    double o = __shfl_xor(v, off);
    v = (o > v) ? o : v;
  }
  return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ inline int wave_or(int v) { return __any(v) ? 1 : 0; }

// Dense solve A x = rhs with partial pivoting, A is [n][ld] row-major with the rhs in column n.
// On return column n holds x.  One wavefront cooperates; returns false on a zero / non-finite pivot.
__device__ inline bool wave_dense_solve(double* __restrict__ A, int n, int ld, int tid) {
  bool ok = true;
  const int tr = tid >> 3, tc = tid & 7;
  for (int k = 0; k < n; ++k) {
    double best = -1.0;
    int bi = k;
    for (int i = k + tid; i < n; i += WAVE) {
      double v = fabs(A[(size_t)i * ld + k]);
      if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      double ob = __shfl_xor(best, off);
      int oi = __shfl_xor(bi, off);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (!(best > 1e-300) || !(best < 1e300)) ok = false;
    if (bi != k) {
      for (int j = k + tid; j <= n; j += WAVE) {
        double a = A[(size_t)k * ld + j], b = A[(size_t)bi * ld + j];
        A[(size_t)k * ld + j] = b;
        A[(size_t)bi * ld + j] = a;
      }
    }
    __syncthreads();
    const double rp = 1.0 / A[(size_t)k * ld + k];
    for (int i = k + 1 + tr; i < n; i += 8) {
      const double m = A[(size_t)i * ld + k] * rp;
      for (int j = k + 1 + tc; j <= n; j += 8) A[(size_t)i * ld + j] -= m * A[(size_t)k * ld + j];
    }
    __syncthreads();
  }
  for (int k = n - 1; k >= 0; --k) {
    const double xk = A[(size_t)k * ld + n] / A[(size_t)k * ld + k];
    for (int i = tid; i < k; i += WAVE) A[(size_t)i * ld + n] -= A[(size_t)i * ld + k] * xk;
    __syncthreads();
    if (tid == 0) A[(size_t)k * ld + n] = xk;
  }
  __syncthreads();
  return ok;
}

// ---------------------------------------------------------------------------------------------------
// K1: element -> compact bus maps, active-bus mask, dense renumbering.  Returns the number of active
// buses (or -1 when it exceeds the capacity nbc).
__device__ inline int build_topology(const GridDev& g, Carve& c, const int* __restrict__ topo,
                                     const int* __restrict__ shunt_bus, unsigned char* __restrict__ status_out,
                                     int nbc, int tid) {
  const int ns = g.n_sub;
  for (int i = tid; i < g.nb_tot; i += WAVE) c.gmap[i] = 0;
  __syncthreads();
  // global bus per element, stored temporarily in the *_c arrays; mark active buses
  for (int l = tid; l < g.n_line; l += WAVE) {
    int bo = topo[g.line_or_pos[l]], be = topo[g.line_ex_pos[l]];
    bool on = (bo >= 1) && (be >= 1);
    int go = on ? g.line_or_sub[l] + (bo - 1) * ns : -1;
    int ge = on ? g.line_ex_sub[l] + (be - 1) * ns : -1;
    c.lor_c[l] = go;
    c.lex_c[l] = ge;
    if (on) { c.gmap[go] = 1; c.gmap[ge] = 1; }
    if (status_out) status_out[l] = on ? 1 : 0;
  }
  for (int i = tid; i < g.n_gen; i += WAVE) {
    int b = topo[g.gen_pos[i]];
    int gb = (b >= 1) ? g.gen_sub[i] + (b - 1) * ns : -1;
    c.gen_c[i] = gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  for (int i = tid; i < g.n_load; i += WAVE) {
    int b = topo[g.load_pos[i]];
    int gb = (b >= 1) ? g.load_sub[i] + (b - 1) * ns : -1;
    c.load_c[i] = gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  for (int i = tid; i < g.n_sto; i += WAVE) {
    int b = topo[g.sto_pos[i]];
    int gb = (b >= 1) ? g.sto_sub[i] + (b - 1) * ns : -1;
    c.sto_c[i] = gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  for (int i = tid; i < g.n_shunt; i += WAVE) {
    int b = shunt_bus[i];
    int gb = (b >= 1) ? g.shunt_sub[i] + (b - 1) * ns : -1;
    c.sh_c[i] = gb;
    if (gb >= 0) c.gmap[gb] = 1;
  }
  __syncthreads();
  // dense renumbering in global-bus order (ballot prefix)
  int base = 0;
  for (int i0 = 0; i0 < g.nb_tot; i0 += WAVE) {
    int i = i0 + tid;
    int act = (i < g.nb_tot) ? c.gmap[i] : 0;
    unsigned long long m = __ballot(act);
    int rank = base + __popcll(m & ((1ull << tid) - 1ull));
    if (i < g.nb_tot) {
      int ci = act ? rank : -1;
      if (act && rank < nbc) c.gid[rank] = i;
      c.gmap[i] = ci;
    }
    base += __popcll(m);
  }
  __syncthreads();
  const int nb = base;
  if (nb > nbc) return -1;
  for (int l = tid; l < g.n_line; l += WAVE) {
    int go = c.lor_c[l], ge = c.lex_c[l];
    c.lor_c[l] = go >= 0 ? c.gmap[go] : -1;
    c.lex_c[l] = ge >= 0 ? c.gmap[ge] : -1;
  }
  for (int i = tid; i < g.n_gen; i += WAVE) { int b = c.gen_c[i]; c.gen_c[i] = b >= 0 ? c.gmap[b] : -1; }
  for (int i = tid; i < g.n_load; i += WAVE) { int b = c.load_c[i]; c.load_c[i] = b >= 0 ? c.gmap[b] : -1; }
  for (int i = tid; i < g.n_sto; i += WAVE) { int b = c.sto_c[i]; c.sto_c[i] = b >= 0 ? c.gmap[b] : -1; }
  for (int i = tid; i < g.n_shunt; i += WAVE) { int b = c.sh_c[i]; c.sh_c[i] = b >= 0 ? c.gmap[b] : -1; }
  __syncthreads();
  return nb;
}

// Pointers read from a parameter block in memory are GENERIC to the compiler: every access becomes a flat_load /
// flat_store, which counts on BOTH vmcnt and lgkmcnt -- an LDS wait then also waits for every result store in flight.
// gptr() re-types such a pointer as global (address space 1) so that global_load / global_store are emitted.
#define GPF_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ GPF_GLOBAL T* gptr(T* p) { return (GPF_GLOBAL T*)p; }

template <int GW = WAVE>
__device__ inline void write_nan_results(const GridDev& g, const Bufs& b, int inst, int tid) {
  auto out = gptr(b.out) + (size_t)inst * g.n_out;
  const float nanv = __builtin_nanf("");
  for (int i = tid; i < g.n_out; i += GW) out[i] = nanv;
  auto to = gptr(b.topo_out) + (size_t)inst * g.dim_topo;
  for (int i = tid; i < g.dim_topo; i += GW) to[i] = -1;
  auto so = gptr(b.shunt_bus_out) + (size_t)inst * g.n_shunt;
  for (int i = tid; i < g.n_shunt; i += GW) so[i] = -1;
  auto ls = gptr(b.line_status) + (size_t)inst * g.n_line;
  for (int i = tid; i < g.n_line; i += GW) ls[i] = 0;
  const double nand = __builtin_nan("");
  auto bvm = gptr(b.bus_vm) + (size_t)inst * g.nb_tot;
  auto bva = gptr(b.bus_va) + (size_t)inst * g.nb_tot;
  for (int i = tid; i < g.nb_tot; i += GW) { bvm[i] = nand; bva[i] = nand; }
}

// ---------------------------------------------------------------------------------------------------
// One complete power flow of one instance.  Returns the GPF_ST_* status (uniform over the wave).
// n_iter_out / nb_out: Newton iterations done / number of active buses.
__device__ inline int solve_instance(const GridDev& g, const Bufs& b, Carve& c, const OutOff& oo, int inst,
                                     int nbc, int nJ, int is_dc, int max_iter, double tol_pu, int tid,
                                     int& n_iter_out, int& nb_out) {
  const double* __restrict__ inj = b.inj + (size_t)inst * g.n_inj;
  const int* __restrict__ topo = b.topo + (size_t)inst * g.dim_topo;
  const int* __restrict__ shb = b.shunt_bus + (size_t)inst * g.n_shunt;
  unsigned char* lstat = b.line_status + (size_t)inst * g.n_line;
  n_iter_out = 0;
  nb_out = 0;

  // ---- K1 ------------------------------------------------------------------------------------------
  const int nb = build_topology(g, c, topo, shb, lstat, nbc, tid);
  if (nb < 0) return 5;
  nb_out = nb;
  const double inv_sn = 1.0 / g.sn_mva;

  // ---- bus types, injections, voltage set-points: thread per bus gathers over the elements (element
  //      order == summation order of the CPU oracle -> deterministic) ---------------------------------------
  for (int ci = tid; ci < nb; ci += WAVE) {
    int bt = BT_PQ;
    double P = 0.0, vs = 1.0;
    for (int i = 0; i < g.n_gen; ++i) {
      if (c.gen_c[i] == ci) {
        if (g.gen_slack[i]) bt = BT_REF;
        else { if (bt != BT_REF) bt = BT_PV; P += inj[oo.inj_gen_p + i] * inv_sn; }
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
    c.Psp[ci] = P - pd * inv_sn;
    c.Qsp[ci] = -qd * inv_sn;
    c.lab[ci] = (bt == BT_REF) ? 1 : 0;
  }
  __syncthreads();
  // unknown numbering: pidx over non-ref buses, qidx over PQ buses (ballot prefix, bus order)
  int npvpq = 0, npq = 0, nref = 0;
  for (int i0 = 0; i0 < nb; i0 += WAVE) {
    int ci = i0 + tid;
    int bt = (ci < nb) ? c.btype[ci] : -1;
    unsigned long long mp = __ballot(bt == BT_PQ || bt == BT_PV);
    unsigned long long mq = __ballot(bt == BT_PQ);
    unsigned long long mr = __ballot(bt == BT_REF);
    unsigned long long below = (1ull << tid) - 1ull;
    if (ci < nb) {
      c.pidx[ci] = (bt == BT_PQ || bt == BT_PV) ? npvpq + __popcll(mp & below) : -1;
      c.qidx[ci] = (bt == BT_PQ) ? npq + __popcll(mq & below) : -1;
    }
    npvpq += __popcll(mp);
    npq += __popcll(mq);
    nref += __popcll(mr);
  }
  __syncthreads();
  if (nref == 0) return 3;
  const int n = npvpq + npq;
  if (n > nJ) return 5;

  // ---- connectivity: label propagation from the reference buses over in-service branches -------------------
  for (int sweep = 0; sweep < nb; ++sweep) {
    int changed = 0;
    for (int l = tid; l < g.n_line; l += WAVE) {
      int f = c.lor_c[l], t = c.lex_c[l];
      if (f >= 0) {
        int lf = c.lab[f], lt = c.lab[t];
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

  // ---- K3: DC solve  B' theta = P - Gs on the non-reference buses --------------------------------------------
  {
    const int nd = npvpq, ld = nd + 1;
    double* A = c.J;
    for (int e = tid; e < nd * ld; e += WAVE) A[e] = 0.0;
    __syncthreads();
    for (int ci = tid; ci < nb; ci += WAVE) {
      const int pi = c.pidx[ci];
      if (pi < 0) continue;
      double diag = 0.0;
      for (int l = 0; l < g.n_line; ++l) {
        const int f = c.lor_c[l], t = c.lex_c[l];
        if (f < 0) continue;
        if (f == ci || t == ci) {
          const double bb = g.br_bdc[l];
          diag += bb;
          const int o = (f == ci) ? t : f;
          if (o != ci) {
            const int po = c.pidx[o];
            if (po >= 0) A[(size_t)pi * ld + po] -= bb;
          } else {
            diag -= bb;  // self loop (both ends on the same bus): contributes nothing
          }
        }
      }
      A[(size_t)pi * ld + pi] += diag;
      A[(size_t)pi * ld + nd] = c.Psp[ci] - c.Gs[ci];
    }
    __syncthreads();
    bool ok = (nd == 0) ? true : wave_dense_solve(A, nd, ld, tid);
    for (int ci = tid; ci < nb; ci += WAVE) {
      const int pi = c.pidx[ci];
      const double th = (pi >= 0) ? A[(size_t)pi * ld + nd] : 0.0;
      c.va[ci] = th;
      c.vm[ci] = (c.btype[ci] == BT_PQ) ? 1.0 : c.vset[ci];
      if (!(fabs(th) < 1e300)) ok = false;
    }
    __syncthreads();
    if (__any(!ok)) return 4;
  }

  // ---- K2: dense Ybus, thread per row gathers over the branches ------------------------------------------------
  if (!is_dc) {
    for (int e = tid; e < 2 * nb * nb; e += WAVE) c.Y[e] = 0.0;
    __syncthreads();
    for (int ci = tid; ci < nb; ci += WAVE) {
      double* Yr = c.Y + (size_t)2 * ci * nb;
      for (int l = 0; l < g.n_line; ++l) {
        const int f = c.lor_c[l], t = c.lex_c[l];
        if (f < 0) continue;
        const double* y = g.br_y + (size_t)8 * l;
        if (f == ci) {
          Yr[2 * f] += y[0]; Yr[2 * f + 1] += y[1];
          Yr[2 * t] += y[2]; Yr[2 * t + 1] += y[3];
        }
        if (t == ci) {
          Yr[2 * f] += y[4]; Yr[2 * f + 1] += y[5];
          Yr[2 * t] += y[6]; Yr[2 * t + 1] += y[7];
        }
      }
      for (int s = 0; s < g.n_shunt; ++s) {
        if (c.sh_c[s] == ci) {
          Yr[2 * ci] += inj[oo.inj_sh_p + s] * g.shunt_fact[s] * inv_sn;
          Yr[2 * ci + 1] -= inj[oo.inj_sh_q + s] * g.shunt_fact[s] * inv_sn;
        }
      }
    }
    __syncthreads();
  }

  int status = 0;
  int it = 0;
  if (!is_dc) {
    // ---- K4/K5: Newton-Raphson ------------------------------------------------------------------------------------
    const int ld = n + 1;
    bool converged = false;
    while (true) {
      // rectangular voltages
      for (int ci = tid; ci < nb; ci += WAVE) {
        double s, co;
        sincos(c.va[ci], &s, &co);
        c.e[ci] = c.vm[ci] * co;
        c.f[ci] = c.vm[ci] * s;
      }
      __syncthreads();
      // S = V conj(Ybus V), mismatch
      double fmax = 0.0;
      bool bad = false;
      for (int ci = tid; ci < nb; ci += WAVE) {
        const double* Yr = c.Y + (size_t)2 * ci * nb;
        double ir = 0.0, ii = 0.0;
        for (int j = 0; j < nb; ++j) {
          const double yr = Yr[2 * j], yi = Yr[2 * j + 1];
          const double ej = c.e[j], fj = c.f[j];
          ir += yr * ej - yi * fj;
          ii += yr * fj + yi * ej;
        }
        const double sr = c.e[ci] * ir + c.f[ci] * ii;
        const double si = c.f[ci] * ir - c.e[ci] * ii;
        c.Sre[ci] = sr;
        c.Sim[ci] = si;
        const int pi = c.pidx[ci], qi = c.qidx[ci];
        if (pi >= 0) {
          const double mp = sr - c.Psp[ci];
          c.J[(size_t)pi * ld + n] = -mp;
          const double a = fabs(mp);
          if (!(a <= 1e300)) bad = true;
          fmax = a > fmax ? a : fmax;
        }
        if (qi >= 0) {
          const double mq = si - c.Qsp[ci];
          c.J[(size_t)(npvpq + qi) * ld + n] = -mq;
          const double a = fabs(mq);
          if (!(a <= 1e300)) bad = true;
          fmax = a > fmax ? a : fmax;
        }
      }
      fmax = wave_max(fmax);
      if (__any(bad)) { status = 1; break; }
      if (fmax < tol_pu) { converged = true; break; }
      if (it >= max_iter) break;
      ++it;
      __syncthreads();
      // Jacobian (dense): every entry is written exactly once
      {
        const int tr = tid >> 3, tc = tid & 7;
        for (int i = tr; i < nb; i += 8) {
          const int pi = c.pidx[i], qi = c.qidx[i];
          if (pi < 0) continue;
          const double ei = c.e[i], fi = c.f[i];
          const double* Yr = c.Y + (size_t)2 * i * nb;
          for (int j = tc; j < nb; j += 8) {
            const int pj = c.pidx[j], qj = c.qidx[j];
            if (pj < 0) continue;
            const double yr = Yr[2 * j], yi = Yr[2 * j + 1];
            const double ej = c.e[j], fj = c.f[j];
            const double a = yr * ej - yi * fj, bq = yr * fj + yi * ej;   // Y_ij V_j
            const double tr_ = ei * a + fi * bq;                           // T = V_i conj(Y_ij V_j)
            const double ti_ = fi * a - ei * bq;
            double dva_r, dva_i, dvm_r, dvm_i;
            const double ivm = 1.0 / c.vm[j];
            if (i == j) {
              const double sr = c.Sre[i], si = c.Sim[i];
              dva_r = ti_ - si;  dva_i = sr - tr_;           // j (S - T)
              dvm_r = (tr_ + sr) * ivm;  dvm_i = (ti_ + si) * ivm;
            } else {
              dva_r = ti_;  dva_i = -tr_;                     // -j T
              dvm_r = tr_ * ivm;  dvm_i = ti_ * ivm;
            }
            c.J[(size_t)pi * ld + pj] = dva_r;
            if (qj >= 0) c.J[(size_t)pi * ld + npvpq + qj] = dvm_r;
            if (qi >= 0) {
              c.J[(size_t)(npvpq + qi) * ld + pj] = dva_i;
              if (qj >= 0) c.J[(size_t)(npvpq + qi) * ld + npvpq + qj] = dvm_i;
            }
          }
        }
      }
      __syncthreads();
      const bool ok = wave_dense_solve(c.J, n, ld, tid);
      if (!ok) { status = 4; break; }
      for (int ci = tid; ci < nb; ci += WAVE) {
        const int pi = c.pidx[ci], qi = c.qidx[ci];
        double va = c.va[ci], vm = c.vm[ci];
        if (pi >= 0) va += c.J[(size_t)pi * ld + n];
        if (qi >= 0) vm += c.J[(size_t)(npvpq + qi) * ld + n];
        if (vm < 0.0) { vm = -vm; va += 3.14159265358979323846; }
        va = remainder(va, 6.28318530717958647692);
        c.va[ci] = va;
        c.vm[ci] = vm;
      }
      __syncthreads();
    }
    if (status == 0 && !converged) status = 1;
  }
  n_iter_out = it;
  if (status != 0) return status;

  // ---- K6: result extraction ------------------------------------------------------------------------------------
  float* out = b.out + (size_t)inst * g.n_out;
  const double sn = g.sn_mva;
  const double RAD2DEG = 57.295779513082320877;
  const double SQRT3 = 1.7320508075688772935;
  if (is_dc) {
    // DC: bus "injection" = sum of branch flows, kept in Sre (MW/sn)
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
    __syncthreads();
  }
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
  // generators: pypower pfsoln (Q split in proportion to the reactive range, slack P = bus balance)
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
  // topo_vect (pandaPowerBackend.py:1489-1524): both ends of an out-of-service line read -1
  int* to = b.topo_out + (size_t)inst * g.dim_topo;
  for (int i = tid; i < g.dim_topo; i += WAVE) { int v = topo[i]; to[i] = v >= 1 ? v : -1; }
  __syncthreads();
  for (int l = tid; l < g.n_line; l += WAVE) {
    if (c.lor_c[l] < 0) { to[g.line_or_pos[l]] = -1; to[g.line_ex_pos[l]] = -1; }
  }
  // float64 bus voltages (pre-cast parity checks, shunt_info / theta of stale buses on the host side)
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
template <bool BIG>
__global__ __launch_bounds__(WAVE) void runpf_kernel(GridDev g, Bufs b, OutOff oo, int lane0, int nbc, int nJ,
                                                     int is_dc, int max_iter, double tol_pu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int inst = lane0 + blockIdx.x;
  const int tid = threadIdx.x;
  Carve c;
  carve_lds(c, smem, g, nbc, nJ, BIG ? b.work + (size_t)inst * b.work_stride : nullptr);
  int n_iter, nb;
  const int st = solve_instance(g, b, c, oo, inst, nbc, nJ, is_dc, max_iter, tol_pu, tid, n_iter, nb);
  __syncthreads();
  if (st != 0) write_nan_results(g, b, inst, tid);
  if (tid == 0) {
    int* s = b.status + (size_t)inst * 4;
    s[0] = st; s[1] = n_iter; s[2] = nb; s[3] = 0;
  }
}

// K9 + K1..K7: one DoNothing environment step per instance.
template <bool BIG>
__global__ __launch_bounds__(WAVE) void step_kernel(GridDev g, Bufs b, OutOff oo, int nbc, int nJ, int max_iter,
                                                    double tol_pu, StepArgs sa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int inst = blockIdx.x;
  const int tid = threadIdx.x;
  Carve c;
  carve_lds(c, smem, g, nbc, nJ, BIG ? b.work + (size_t)inst * b.work_stride : nullptr);
  // ---- K9: chronics row -> injections (float32 API values, exactly as _BackendAction stores them) ----------------
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
    for (int i = tid; i < g.n_gen; i += WAVE) {
      if (!g.gen_slack[i]) sum_prod += (double)ch[2 * g.n_load + i];
    }
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
      inj[oo.inj_gen_vm + i] = (double)(pv_kv / vn);   // float32 division, as pandaPowerBackend.py:927
    }
    __syncthreads();
  }
  int n_iter = 0, nb = 0, st = 0, rounds = 0;
  int* ovc = b.overflow_count + (size_t)inst * g.n_line;   // env._protection_counter (baseEnv.py:3367-3370)
  int* dround = b.disc_round + (size_t)inst * g.n_line;    // backend._disconnected_during_cf
  float* rho = b.rho + (size_t)inst * g.n_line;
  float* out = b.out + (size_t)inst * g.n_out;
  int* topo = b.topo + (size_t)inst * g.dim_topo;
  constexpr int MAXK = 4;                                    // n_line <= 256 (checked on the host)
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
    st = solve_instance(g, b, c, oo, inst, nbc, nJ, sa.is_dc, max_iter, tol_pu, tid, n_iter, nb);
    __syncthreads();
    if (st != 0 || !sa.cascade || rounds >= sa.max_rounds) break;   // at most max_rounds re-solves
    // K7: Backend.next_grid_state (grid2op/Backend/backend.py:1476-1520)
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
  if (st != 0) write_nan_results(g, b, inst, tid);
  __syncthreads();
  // env bookkeeping after the step (grid2op/Environment/baseEnv.py:3346-3370)
  for (int l = tid; l < g.n_line; l += WAVE) {
    const float lim = b.thermal_limit[l];
    const float a = out[oo.a_or + l];
    rho[l] = a / lim;                                         // backend.py:1145-1168 (NaN when diverged)
    if (a > sa.soft_overflow * lim) ovc[l] += 1; else ovc[l] = 0;
  }
  if (tid == 0) {
    int* s = b.status + (size_t)inst * 4;
    s[0] = st; s[1] = n_iter; s[2] = nb; s[3] = rounds;
  }
}

// device-side lane utilities ------------------------------------------------------------------------------------------
__global__ void fanout_kernel(GridDev g, Bufs b, int src, int dst0, int n_dst, const int* __restrict__ out_lines) {
  const int k = blockIdx.x;
  if (k >= n_dst) return;
  const int dst = dst0 + k;
  const int tid = threadIdx.x;
  const double* sinj = b.inj + (size_t)src * g.n_inj;
  double* dinj = b.inj + (size_t)dst * g.n_inj;
  for (int i = tid; i < g.n_inj; i += blockDim.x) dinj[i] = sinj[i];
  const int* st = b.topo + (size_t)src * g.dim_topo;
  int* dt = b.topo + (size_t)dst * g.dim_topo;
  const int ol = out_lines ? out_lines[k] : -1;
  const int po = (ol >= 0 && ol < g.n_line) ? g.line_or_pos[ol] : -1;
  const int pe = (ol >= 0 && ol < g.n_line) ? g.line_ex_pos[ol] : -1;
  for (int i = tid; i < g.dim_topo; i += blockDim.x) dt[i] = (i == po || i == pe) ? -1 : st[i];
  const int* ss = b.shunt_bus + (size_t)src * g.n_shunt;
  int* ds = b.shunt_bus + (size_t)dst * g.n_shunt;
  for (int i = tid; i < g.n_shunt; i += blockDim.x) ds[i] = ss[i];
}

}  // namespace gpf
