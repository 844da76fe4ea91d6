// gridpf_common.hpp -- definitions shared by the device code and the host side of libgridpf.so (gfx950 / CDNA4 only).
//
// Mapping: one wavefront (or 1/2, 1/4, 2 of them) per grid instance ("lane" in the C ABI; called "instance" in the device
// code to avoid the clash with SIMD lanes); all per-instance state lives in LDS, the per-instance input and output rows
// are instance-major so that the lanes of a wavefront touch consecutive addresses.
//
// Pipeline per instance (reference counterparts, paths relative to the reference checkout):
//   K1 topology compaction      PandaPowerBackend.apply_action bus scatter + pandapower pd2ppc bus lookup
//                               (grid2op/Backend/pandaPowerBackend.py:920-975)
//   K2 Ybus assembly            pandapower makeYbus (SURVEY.md A4')
//   K3 DC solve                 runpp(init="dc") / rundcpp  (pandaPowerBackend.py:1086-1090)
//   K4 mismatch + Jacobian      pypower newtonpf / dSbus_dV
//   K5 block-sparse LU + solve  scipy.sparse.linalg.spsolve per Newton iteration (:1081-1083)
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
  double inv_sn_mva;       // 1 / sn_mva (host, correctly rounded: what the device's own division gave)
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

// Field lists for the grid-specialised (run-time compiled) kernels, gridpf_capi.hip: jit_header().  Every scalar of the launch
// parameter block that depends on the GRID alone is listed here; pointers and per-launch values stay run-time values.
#define GPF_GRIDDEV_INTS(X) X(n_sub) X(n_busbar) X(nb_tot) X(n_line) X(n_gen) X(n_load) X(n_sto) X(n_shunt) X(dim_topo) X(n_inj) X(n_out) X(n_chron)
#define GPF_GRIDDEV_DBLS(X) X(sn_mva) X(inv_sn_mva)

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
  // BaseEnv._times_before_line_status_actionable (obs.time_before_cooldown_line) of a DoNothing step (baseEnv.py:3352-3358, 2590-2597):
  // decremented every step, set to NB_TIMESTEP_RECONNECTION when the protections trip the line, raised to the remaining duration of a
  // maintenance / hazard under way.  Maintained by the step kernel when StepArgs::nb_ts_reco >= 0.
  int* cooldown;               // [B][n_line]
  const unsigned short* maint_dur;   // [n_tab][T][n_line] steps the maintenance / hazard under way at that row still lasts (0: none), or nullptr
  short* traj_cool;            // [traj_cap][B][n_line] cooldown of every step of the last multi-step launch (with traj_rho), or nullptr
  const float* lane_gen_delta; // [B][n_gen] MW added to prod_p after the chronics (redispatch, baseEnv.py:2211-2470), or nullptr
  const unsigned char* maint;  // [n_tab][T][n_line] 1: the line is in maintenance at that chronics row (forced out of service), or nullptr
  const int* topo0;            // [B][dim_topo] topology last SENT by the host (what an auto-reset restores)
  unsigned char* done;         // [B] 1: the lane's last step ended its episode (power flow diverged / grid islanded)
  int* episode;                // [B][2] {steps survived since the last reset, number of auto-resets}
  float* traj_rho;             // [traj_cap][B][n_line] rho of every step of the last multi-step launch, or nullptr
  signed char* traj_status;    // [traj_cap][B] GPF_ST_* of every step of the last multi-step launch
  // complete backend observation of EVERY step of a multi-step launch (gpf_set_trajectory(.., GPF_TRAJ_OBS)), or nullptr: the
  // step kernel then writes each step's results row / topo_vect / shunt buses / line status into row [step][lane] of these
  // buffers instead of overwriting the lane's single row, and copies the last step's rows to out / topo_out / ... at the end
  float* traj_out;             // [traj_cap][B][n_out]
  int* traj_topo;              // [traj_cap][B][dim_topo]
  int* traj_shb;               // [traj_cap][B][n_shunt]
  unsigned char* traj_lstat;   // [traj_cap][B][n_line]
  int traj_cap;
  long long lane_stride;       // B (padded lane count): stride of the trajectory buffers
  long long n_real_lanes;      // lanes >= this index are padding (ghost lanes of instance groups / lane lists)
};

// Injection dynamics of the environment (opt-in, gpf_set_env_dynamics): what BaseEnv.step does to the generator / storage
// set-points between the chronics and the backend -- storage state of charge (baseEnv.py:2829-2905, 2777-2790), accumulation of
// the agents' redispatch (:2101-2115), the _make_redisp gate (:2188-2209) and the ramp-limited projection
// (_compute_dispatch_vect :2211-2470, the exact separable-QP solution of gridpf_redispatch.hpp) -- evaluated by the step kernel at
// every step of a launch.  State arrays are float32 like the reference's (dt_float).
struct EnvDyn {
  int on;                        // 0: off (every pointer below may be null)
  int hold_storage;              // the storage action is applied at every step of the launch (else at its first step only)
  int loss_on;                   // Parameters.ACTIVATE_STORAGE_LOSS
  double coeff;                  // delta_time_seconds / 3600
  double eps_poly, tol_poly;     // BaseEnv._epsilon_poly / _tol_poly
  // per-lane state [B][n_gen] / [B][n_storage] / [B]
  float *target, *actual, *prev_p;      // _target_dispatch, _actual_dispatch, _gen_activeprod_t_redisp
  unsigned char* already;               // _already_modified_gen
  float* charge;                        // _storage_current_charge (MWh)
  float* amount_prev;                   // _amount_storage_prev
  float* limit;                         // [B][n_gen] _limit_curtailment (ratio of pmax, 1 = not curtailed)
  float* curt_prev;                     // [B] _sum_curtailment_mw_prev
  unsigned char* fresh;                 // [B] 1: no step since the reset (nb_time_step == 0: prev_p := the step's own set-points)
  int* illegal;                         // [B] steps since the reset whose action was cancelled as an illegal redispatch (info["is_illegal_redisp"])
  // per-lane actions of the NEXT launch [B][n_gen] / [B][n_storage]: redispatch is consumed by the first step
  const float *act_redisp, *act_storage;
  const float* act_curtail;             // [B][n_gen] curtailment action (ratio of pmax; -1 = no change), consumed by the first step
  const unsigned char* renewable;       // [n_gen] gen_renewable (null: no curtailment)
  // characteristics [n_gen] / [n_storage]
  const double *pmin, *pmax, *ramp_up, *ramp_down;
  const unsigned char* redispatchable;
  const double *Emax, *Emin, *loss, *eff_c, *eff_d;
  const float* charge0;                 // [n_storage] state of charge after a reset
};

// Topology-derived state of the grid's REFERENCE topology, shared by every lane that is on it, for ONE-STEP launches (gpf_step_n with
// n_steps = 1: agents that act at every step).  What a later step of a multi-step launch finds in LDS / registers -- element -> bus maps, bus
// types, the Ybus blocks, the factored DC matrix, the verdicts of K1 / connectivity -- is what a one-step launch rebuilds from the lane's
// topology row every time (20-25 % of its cycles, tools/phase_timing.py).  The blob holds that state for ONE key: the topology row, shunt
// buses and shunt set-points of the engine's pristine lane (the state every lane starts from and an agent that only moves injections never
// leaves).  A lane whose rows equal the key loads the blob -- 2-19 KB that every block reads, so they come from L2 -- and runs its step like a
// step whose topology stands (SolveCtl::reuse); any other lane rebuilds as before.  (A blob PER LANE was measured first: the burst of
// 1 024 x 19 KB at the start of a launch cost as much as the rebuild it saved.)  The state is written once, by the first lane on the key that
// wins the claim in header word 0, and trusted only by LATER launches (the word carries the writer's launch number).
// Layout: [16 ints header][key ints: dim_topo + n_shunt][off_kd: 2 n_shunt doubles][off_m: n_m ints = the LDS range btype .. sub_bb][off_y: Ybus,
// (2 n_up + n_bus) double2 of the register layout (YR kernels: second blob) or nslot_y double2][off_d: nslot doubles DC factors]
struct KeepArgs {
  unsigned char* p;        // [2][stride]: blob of the kernels with Ybus in LDS, blob of the kernels with Ybus in registers; nullptr: off
  long long stride;        // bytes per blob (multiple of 16)
  int launch;              // number of this launch (>= 1): a state written by launch k carries KEEP_VALID + k in header word 0
  int off_kd, off_m, off_y, off_d, n_m;
};
constexpr int KEEP_HDR_INTS = 16;      // [0] KEEP_KEYED / KEEP_CLAIMED / KEEP_VALID + launch, [1] status | nb << 8, [2] dc_base | gen_base << 1 | DC factors held << 2 | (dc_out + 1) << 3
constexpr int KEEP_KEYED = 1, KEEP_CLAIMED = 2, KEEP_VALID = 16;

struct StepArgs {
  int t, T, rebalance_on, cascade, nb_ts_allowed, max_rounds, is_dc;
  int lane0;         // first lane of a contiguous launch (gpf_simulate_batch steps a sub-range; 0 for gpf_step_n)
  int n_steps;       // env steps per launch (t, t+1, ...): lane state and topology-derived tables stay in LDS in between
  int warm_start;    // opt-in: Newton starts from the previous step's voltages while the topology stands (see SolveCtl::warm)
  int auto_reset;    // a lane whose step failed restarts from the topology last sent by the host, counters cleared
  int nb_ts_reco;    // Parameters.NB_TIMESTEP_RECONNECTION: cooldown a line gets when the protections trip it; < 0: the cooldown counters are not maintained
  double rebalance;
  float hard_overflow, soft_overflow;
  KeepArgs keep;     // (gpf_step_n, n_steps = 1, single-busbar kernels without topology classes)
};

// results-row offsets (must match gpf_layout in include/gridpf.h)
struct OutOff {
  int p_or, q_or, v_or, a_or, th_or, p_ex, q_ex, v_ex, a_ex, th_ex;
  int gen_p, gen_q, gen_v, gen_th, load_p, load_q, load_v, load_th, sto_p, sto_q, sto_v, sto_th, sh_p, sh_q, sh_v;
  int inj_gen_p, inj_gen_vm, inj_load_p, inj_load_q, inj_sto_p, inj_sto_q, inj_sh_p, inj_sh_q;
};

#define GPF_OUTOFF_INTS(X) X(p_or) X(q_or) X(v_or) X(a_or) X(th_or) X(p_ex) X(q_ex) X(v_ex) X(a_ex) X(th_ex) X(gen_p) X(gen_q) X(gen_v) X(gen_th) \
  X(load_p) X(load_q) X(load_v) X(load_th) X(sto_p) X(sto_q) X(sto_v) X(sto_th) X(sh_p) X(sh_q) X(sh_v) X(inj_gen_p) X(inj_gen_vm) X(inj_load_p) \
  X(inj_load_q) X(inj_sto_p) X(inj_sto_q) X(inj_sh_p) X(inj_sh_q)
#define GPF_COUNT_FIELD(f) +1
static_assert(sizeof(OutOff) == sizeof(int) * (0 GPF_OUTOFF_INTS(GPF_COUNT_FIELD)), "GPF_OUTOFF_INTS must list every field of OutOff");

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

typedef signed char i8;


// Pointers read from a parameter block in memory are GENERIC to the compiler: every access becomes a flat_load /
// flat_store, which counts on BOTH vmcnt and lgkmcnt -- an LDS wait then also waits for every result store in flight.
// gptr() re-types such a pointer as global (address space 1) so that global_load / global_store are emitted.
#define GPF_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ GPF_GLOBAL T* gptr(T* p) { return (GPF_GLOBAL T*)p; }

// Rows the results of one solve go to: the lane's own row of out / topo_out / shunt_bus_out / line_status (orow = lane), or row
// orow = step * lane_stride + lane of the per-step observation trajectory (otraj).
template <int GW = WAVE>
__device__ inline void write_nan_results(const GridDev& g, const Bufs& b, int inst, int tid, int orow, bool otraj) {
  auto out = gptr(otraj ? b.traj_out : b.out) + (size_t)orow * g.n_out;
  const float nanv = __builtin_nanf("");
  for (int i = tid; i < g.n_out; i += GW) out[i] = nanv;
  auto to = gptr(otraj ? b.traj_topo : b.topo_out) + (size_t)orow * g.dim_topo;
  for (int i = tid; i < g.dim_topo; i += GW) to[i] = -1;
  auto so = gptr(otraj ? b.traj_shb : b.shunt_bus_out) + (size_t)orow * g.n_shunt;
  for (int i = tid; i < g.n_shunt; i += GW) so[i] = -1;
  auto ls = gptr(otraj ? b.traj_lstat : b.line_status) + (size_t)orow * g.n_line;
  for (int i = tid; i < g.n_line; i += GW) ls[i] = 0;
  const double nand = __builtin_nan("");
  auto bvm = gptr(b.bus_vm) + (size_t)inst * g.nb_tot;
  auto bva = gptr(b.bus_va) + (size_t)inst * g.nb_tot;
  for (int i = tid; i < g.nb_tot; i += GW) { bvm[i] = nand; bva[i] = nand; }
}

// ---------------------------------------------------------------------------------------------------
// One complete power flow of one instance.  Returns the GPF_ST_* status (uniform over the wave).

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

}  // namespace gpf
