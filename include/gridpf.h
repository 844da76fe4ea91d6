/* gridpf.h -- C ABI of libgridpf.so: the MI355X-native batched power-flow engine behind the
 * grid2op Backend plugin surface.
 *
 * The reference (Grid2op/grid2op, 100 % Python) has no FFI of its own: its Backend plugin calls the
 * third-party package pandapower.  This header declares the entry points a grid2op Backend binds
 * instead (via ctypes, see INTEGRATION.md and grid2op_amd/_capi.py); each one cites the reference
 * interface it replaces.  Paths are relative to the reference checkout.
 *
 * Conventions: every function returns 0 on success and a negative GPF_E_* code on failure (message via
 * gpf_last_error()); no exception crosses the ABI; the caller owns every host buffer; a handle is
 * thread-compatible (use one handle per host thread); all work of a handle is queued on ONE HIP
 * stream owned by the handle and the gpf_get_* calls synchronise that stream.
 *
 * A "lane" is one independent grid instance (one environment copy or one N-1 contingency).  All
 * per-lane buffers are lane-major: row `lane` of a [n_lanes][stride] array.
 *
 * Units (same as grid2op/Backend/backend.py:563-760): MW, MVAr, kV, A, degrees; buses are LOCAL bus
 * ids 1..n_busbar, -1 = disconnected; generator voltage setpoints are in per unit.
 */
#ifndef GRIDPF_H
#define GRIDPF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPF_OK 0
#define GPF_E_INVALID (-1)   /* bad argument                                  */
#define GPF_E_DEVICE (-2)    /* HIP runtime error (no device, OOM, launch)    */
#define GPF_E_CAPACITY (-3)  /* grid too large for the compiled kernels       */
#define GPF_E_UNSUPPORTED (-4) /* optional facility not available on this host (gpf_jit_enable without hipcc) */

/* per-lane solver status written by gpf_runpf / gpf_step (status[lane*4 + 0]) */
#define GPF_ST_CONVERGED 0
#define GPF_ST_MAXITER 1     /* Newton did not reach the tolerance in max_iter iterations */
#define GPF_ST_ISLANDED 2    /* an active bus is not connected to a reference bus         */
#define GPF_ST_NOSLACK 3     /* no in-service slack generator                             */
#define GPF_ST_SINGULAR 4    /* zero / non finite pivot                                   */
#define GPF_ST_CAPACITY 5    /* more active buses than the handle was sized for           */
#define GPF_ST_NOTRUN (-1)

#define GPF_MAX_BUSBAR 64          /* busbars per substation a grid may have (the reference takes any n_busbar_per_sub,
                                      pandaPowerBackend.py:562-577; grid2op/tests/test_issue_l2g_128.py:218 uses 6)                  */
#define GPF_MAX_BUSBAR_BLOCKS 3    /* ... of which the NB = n_busbar block kernels cover 1..3.  Lanes with split substations normally
                                      run the single-busbar kernel on the bus-level graph of their topology class (any busbar count);
                                      only the developer fallback GRIDPF_NO_CLASSES=1 / a class that cannot be built needs the block
                                      kernels and is refused beyond 3 busbars (GPF_E_CAPACITY, at launch planning)                   */

typedef struct gpf_engine* gpf_handle;

/* Static description of one grid = what PandaPowerBackend.load_grid + _init_private_attrs derive
 * from the grid file (grid2op/Backend/pandaPowerBackend.py:356-617, 670-874) plus the per-unit
 * branch model pandapower's pd2ppc/makeYbus builds.  Arrays are copied by gpf_create.            */
typedef struct gpf_grid_desc {
  int32_t n_sub, n_busbar;                 /* global bus = sub + (local-1)*n_sub (GridObjects.py:4683) */
  int32_t n_line, n_gen, n_load, n_storage, n_shunt, dim_topo;
  double sn_mva;
  const double* sub_vn_kv;                 /* [n_sub]                                              */
  const int32_t* line_or_sub;              /* [n_line] lines first, then trafos (:462-471)         */
  const int32_t* line_ex_sub;
  const int32_t* line_or_pos_topo_vect;    /* [n_line] positions in topo_vect (GridObjects.py:1409) */
  const int32_t* line_ex_pos_topo_vect;
  const double* br_y;                      /* [n_line][8] yff.re,yff.im,yft.re,yft.im,ytf.re,ytf.im,ytt.re,ytt.im (pu) */
  const double* br_bdc;                    /* [n_line] 1/(x*ratio) DC susceptance (pu)             */
  const int32_t* gen_sub;                  /* [n_gen]                                              */
  const int32_t* gen_pos_topo_vect;
  const double* gen_min_q;                 /* [n_gen] MVAr, used for the Q split of co-located gens */
  const double* gen_max_q;
  const uint8_t* gen_slack;                /* [n_gen] 1 = slack generator (reference bus follows it) */
  const int32_t* load_sub;                 /* [n_load]                                             */
  const int32_t* load_pos_topo_vect;
  const int32_t* storage_sub;              /* [n_storage]                                          */
  const int32_t* storage_pos_topo_vect;
  const int32_t* shunt_sub;                /* [n_shunt]                                            */
  const double* shunt_fact;                /* [n_shunt] step*(vn_bus/vn_shunt)^2                   */
  /* pristine lane state (what reset() restores, pandaPowerBackend.py:334-354, 872) */
  const double* init_inj;                  /* [n_inj] see gpf_layout                               */
  const int32_t* init_topo;                /* [dim_topo]                                           */
  const int32_t* init_shunt_bus;           /* [n_shunt]                                            */
} gpf_grid_desc;

/* Row layouts of the per-lane buffers (offsets in elements). */
typedef struct gpf_layout {
  /* injections row (double): what apply_action scatters (pandaPowerBackend.py:925-969) */
  int32_t n_inj;
  int32_t inj_gen_p, inj_gen_vm, inj_load_p, inj_load_q, inj_storage_p, inj_storage_q, inj_shunt_p, inj_shunt_q;
  /* results row (float = grid2op dt_float): what _fetch_data_pf_converged gathers (:1122-1218) */
  int32_t n_out;
  int32_t out_p_or, out_q_or, out_v_or, out_a_or, out_theta_or;
  int32_t out_p_ex, out_q_ex, out_v_ex, out_a_ex, out_theta_ex;
  int32_t out_gen_p, out_gen_q, out_gen_v, out_gen_theta;
  int32_t out_load_p, out_load_q, out_load_v, out_load_theta;
  int32_t out_storage_p, out_storage_q, out_storage_v, out_storage_theta;
  int32_t out_shunt_p, out_shunt_q, out_shunt_v;
  /* chronics row (float): load_p[n_load], load_q[n_load], prod_p[n_gen], prod_v[n_gen] (kV) */
  int32_t n_chron;
  int32_t chron_load_p, chron_load_q, chron_prod_p, chron_prod_v;
  int32_t nb_total;                        /* n_sub*n_busbar: row length of the bus voltage buffers */
} gpf_layout;

const char* gpf_last_error(void);
/* ABI version of the library = GPF_ABI_VERSION of the header it was built from.  A binding MUST compare the two before any other
 * call (grid2op_amd/_capi.py does): 300 = round 4 (gpf_set_trajectory(h, cap, what), 22 device pointers, GPF_ST_REDISPATCH,
 * gpf_device_pointers_n); 310 = + gpf_jit_*, GPF_E_UNSUPPORTED, gpf_set_profiling mode 3; 321 = 28 device pointers (action buffers and
 * dispatch / charge state of the environment dynamics), gpf_lane_actions_on_device; 322 = + gpf_get_results_pinned; 323 = gpf_step_opts::track_cooldown,
 * GPF_DEVICE_NONE (header-only handles). */
#define GPF_ABI_VERSION 323
int gpf_version(void);
/* Bitwise run-to-run reproducibility is the DEFAULT on every grid: the same lane inputs give bit-identical results from run to
 * run and whatever the lane's position in the batch (grid2op's determinism contract: same seeds -> same episode,
 * grid2op/Environment/baseEnv.py seed()).  Small grids are solved by one wavefront per lane (the LDS applies the atomics of a
 * wavefront in a fixed order); grids with >= 64 substations by 2 wavefronts per lane whose accumulations are arranged so that no
 * sum depends on the interleaving of the two (accumulation loops on wavefront 0, every destination of an LU pass owned by one
 * wavefront, per-wavefront partial sums added in a fixed order).  flag != 0 additionally forces ONE wavefront per lane on the
 * large grids (~20 % slower there; kept for cross-checks -- its results differ from the 2-wavefront kernel's in the last bits,
 * each variant being reproducible in itself). */
int gpf_set_deterministic(gpf_handle h, int32_t flag);
/* Number of HIP devices visible to this process (0 and GPF_OK when there is none): what a single-process caller
 * shards its lane batch over (grid2op_amd/sharding.py ShardedEngine; the reference's own parallelism is one process per
 * environment, Runner/runner.py:1071-1253, Environment/baseMultiProcessEnv.py:293). */
int gpf_device_count(int32_t* n_devices);

/* load_grid (pandaPowerBackend.py:356): build an engine with `n_lanes` lanes on HIP device `device`.
 * Every lane starts in the pristine state.
 * device = GPF_DEVICE_NONE: a HEADER-ONLY handle -- every host-side step of the construction runs (validation, symbolic analysis of the
 * substation graph, static tables, launch planning for `n_lanes` lanes) but nothing is allocated on a device, and none is needed: what
 * works on it is gpf_jit_source (the header the grid-specialised kernels are compiled with -- pure grid arithmetic), gpf_get_plan (the
 * kernel variant a launch over all lanes would take), gpf_get_layout, gpf_n_lanes and gpf_destroy; every other entry point fails with
 * GPF_E_DEVICE.  It is how the ahead-of-time objects of a NEW grid are prepared on a machine without a GPU
 * (python -m grid2op_amd.aot). */
#define GPF_DEVICE_NONE (-1)
int gpf_create(const gpf_grid_desc* desc, int32_t n_lanes, int32_t device, gpf_handle* out);
/* close (pandaPowerBackend.py:1411-1423) */
int gpf_destroy(gpf_handle h);
int gpf_get_layout(gpf_handle h, gpf_layout* out);
int gpf_n_lanes(gpf_handle h);
/* Lane capacity of the per-lane device buffers (n_lanes rounded up, plus padding lanes the kernels use for instance groups). */
int gpf_lane_capacity(gpf_handle h);

/* apply_action, injection half (pandaPowerBackend.py:925-969): overwrite rows lane0..lane0+n-1.
 * inj is [n][n_inj] double. */
int gpf_set_injections(gpf_handle h, int32_t lane0, int32_t n, const double* inj);
/* apply_action, topology half (:920-922, 941-951, 964-975): topo is [n][dim_topo] local bus ids
 * (the layout of _BackendAction.current_topo.values), shunt_bus is [n][n_shunt] (may be NULL). */
int gpf_set_topology(gpf_handle h, int32_t lane0, int32_t n, const int32_t* topo, const int32_t* shunt_bus);
int gpf_get_injections(gpf_handle h, int32_t lane0, int32_t n, double* inj);
int gpf_get_topology(gpf_handle h, int32_t lane0, int32_t n, int32_t* topo, int32_t* shunt_bus);
/* _disconnect_line (:1464-1475): both ends of line `line_id` of lane `lane` go to -1. */
int gpf_disconnect_line(gpf_handle h, int32_t lane, int32_t line_id);
/* reset (:334-354): back to the pristine state. */
int gpf_reset_lanes(gpf_handle h, int32_t lane0, int32_t n);
/* copy (:1289-1409): device-side copy of the complete state (inputs and last results; with gpf_set_env_dynamics on also the
 * dispatch / storage / curtailment state of the environment's injection dynamics) of `n` lanes. */
int gpf_copy_lanes(gpf_handle h, int32_t src_lane0, int32_t dst_lane0, int32_t n);
/* N-1 fan-out (Reward/n1Reward.py:75-99, Observation/_obsEnv.py:321-503): lanes dst0+k (k < n_out)
 * become copies of `src_lane` (injections, topology, shunts; with gpf_set_env_dynamics on also its dispatch / storage /
 * curtailment state) with line out_lines[k] disconnected (out_lines[k] < 0: plain copy). */
int gpf_fanout_n1(gpf_handle h, int32_t src_lane, int32_t dst_lane0, int32_t n_out, const int32_t* out_lines);

/* runpf (pandaPowerBackend.py:1220-1255 -> pp.runpp / pp.rundcpp :1078-1120): one AC Newton-Raphson
 * (init="dc", <= max_iter iterations, ||F||inf < tol_mva/sn_mva) or DC power flow per lane, then the
 * result extraction of _fetch_data_pf_converged.  Asynchronous (queued on the handle's stream). */
int gpf_runpf(gpf_handle h, int32_t lane0, int32_t n, int32_t is_dc, int32_t max_iter, double tol_mva);

/* The single-environment plugin path in ONE call (what HipBackend.runpf does per power flow): gpf_set_injections +
 * gpf_set_topology + gpf_runpf + gpf_get_results for one lane through a device-mapped pinned host block (one dispatch
 * reads the inputs from it into the lane's rows, one writes the result rows back into it; no staged copies), synchronised once
 * (apply_action pandaPowerBackend.py:902-975, runpf :1220-1255, getters :1566-1619).  inj [n_inj], topo [dim_topo],
 * shunt_bus [n_shunt] (required when the grid has shunts); output pointers as in gpf_get_results (any may be NULL).
 * Synchronous. */
int gpf_solve_lane(gpf_handle h, int32_t lane, const double* inj, const int32_t* topo, const int32_t* shunt_bus, int32_t is_dc,
                   int32_t max_iter, double tol_mva, float* out, int32_t* topo_vect, int32_t* shunt_bus_out, uint8_t* line_status,
                   int32_t* status, double* bus_vm, double* bus_va);

/* gpf_get_results without the second host copy: the rows asked for (bit k of `what`: 0 out, 1 topo_vect, 2 shunt_bus, 3 line_status,
 * 4 status, 5 bus_vm, 6 bus_va, 7 rho [n][n_line] float32) are copied by DMA into a pinned block the engine owns and ptrs[k] (k < 8)
 * points at piece k inside it (NULL: not asked for) -- valid until the next call of this function.  A host agent that reads every
 * lane's rows at every step gets the PCIe rate instead of the pageable-copy rate.  Synchronous. */
int gpf_get_results_pinned(gpf_handle h, int32_t lane0, int32_t n, int32_t what, void** ptrs /* [8] */);

/* Getters (pandaPowerBackend.py:1566-1619, 278-301, 1439-1462, 1486): synchronise, then copy rows
 * lane0..lane0+n-1.  Any pointer may be NULL.
 *   out         [n][n_out]    float   (NaN everywhere when the lane did not converge, :1257-1287)
 *   topo_vect   [n][dim_topo] int32   (-1 everywhere when not converged)
 *   shunt_bus   [n][n_shunt]  int32
 *   line_status [n][n_line]   uint8
 *   status      [n][4]        int32   {GPF_ST_*, n_iter, n_active_bus, n_cascade_rounds}
 *   bus_vm/va   [n][nb_total] double  pu / degrees, NaN for inactive buses (pre-cast parity checks) */
int gpf_get_results(gpf_handle h, int32_t lane0, int32_t n, float* out, int32_t* topo_vect, int32_t* shunt_bus,
                    uint8_t* line_status, int32_t* status, double* bus_vm, double* bus_va);

/* ---- batched environment stepping (device-resident chronics) ----------------
 * chronics: [n_tables][T][n_chron] float, resident in HBM.  Lane k reads row
 * (t + lane_offset[k]) mod T of table lane_table[k]; loads are multiplied by lane_scale[k][0..2*n_load)
 * (NULL = 1); if `rebalance` != 0 the non-slack prod_p are rescaled so that sum(prod_p) =
 * rebalance * sum(load_p) (Environment/baseEnv.py:2516-2563 feeds these 4 vectors each step). */
int gpf_upload_chronics(gpf_handle h, int32_t n_tables, int32_t T, const float* data);
/* Scheduled maintenance of the uploaded chronics tables: [n_tables][T][n_line] uint8, 1 = the line is in maintenance at that
 * row (maintenance.csv, grid2op/Chronics/gridStateFromFile.py:520-600 -> the "maintenance" modification the environment applies
 * every step, Environment/baseEnv.py:2516-2563): gpf_step / gpf_step_n force such a line out of service; it stays out afterwards
 * (nothing reconnects it for a DoNothing agent).  NULL removes the table. */
int gpf_upload_maintenance(gpf_handle h, int32_t n_tables, int32_t T, const uint8_t* data);
/* Hazards of the uploaded chronics tables (hazards.csv, Chronics/gridStateFromFile.py:478-490: unplanned outages), same shape and
 * same effect on the backend as the maintenance table -- the line is forced out of service at the rows where it is flagged (the
 * "hazards" modification the environment applies every step); the two tables are independent, the device applies their union. */
int gpf_upload_hazards(gpf_handle h, int32_t n_tables, int32_t T, const uint8_t* data);
/* Remaining duration (steps, incl. the current one) of the maintenance / hazard under way at every row, [n_tables][T][n_line] uint16: what the
 * line cooldowns are raised to during an outage (gpf_step_opts::track_cooldown; GridValue.get_maintenance_duration_1d / get_hazard_duration_1d,
 * grid2op/Chronics/gridValue.py:339).  By default the library derives it from the uploaded outage tables (one backward scan); a caller whose
 * tables are a WINDOW of longer chronics passes the true values here, because an outage that runs past the end of the window looks shorter than it
 * is.  Used where an outage table flags the line; NULL: back to the derived values. */
int gpf_upload_outage_durations(gpf_handle h, int32_t n_tables, int32_t T, const uint16_t* data);
int gpf_set_lane_chronics(gpf_handle h, const int32_t* lane_table, const int32_t* lane_offset, const float* lane_scale);
int gpf_set_thermal_limits(gpf_handle h, const float* limit_a /* [n_line] */);
/* One DoNothing env.step for every lane (Environment/baseEnv.py:3562 -> Backend.next_grid_state
 * backend.py:1433-1521): chronics row -> injections -> AC power flow -> results -> overflow counters;
 * when `cascade` != 0 lines above hard_overflow*limit (or soft-overflowed for more than nb_ts_allowed
 * steps) are tripped and the power flow re-run, at most max_rounds times.  is_dc != 0 runs the DC power flow
 * instead (Parameters.ENV_DC, grid2op/Parameters.py:273 -> runpf(is_dc=True)).  Asynchronous. */
int gpf_step(gpf_handle h, int32_t t, int32_t max_iter, double tol_mva, double rebalance, int32_t cascade,
             float hard_overflow, float soft_overflow, int32_t nb_ts_allowed, int32_t max_rounds, int32_t is_dc);
/* Options of gpf_step_n (the arguments of gpf_step, plus auto_reset). */
typedef struct gpf_step_opts {
  int32_t max_iter;
  double tol_mva;
  double rebalance;
  int32_t cascade;
  float hard_overflow, soft_overflow;
  int32_t nb_ts_allowed, max_rounds;
  int32_t is_dc;
  int32_t auto_reset;    /* != 0: a lane whose step fails (game over: BaseEnv.step sets done when the backend diverges,
                            Environment/baseEnv.py:3847-3931) restarts at the next step from the topology the host sent last
                            (gpf_set_topology / gpf_reset_lanes), overflow counters cleared -- what env.reset() does to the
                            backend (Environment/environment.py:1418 reset_grid); the chronics cursor keeps running */
  int32_t warm_start;    /* != 0 (OPT-IN, NOT what PandaPowerBackend does): steps 2..n of a launch start Newton from the previous
                            step's voltages while the lane's topology stands, instead of the DC initialisation pandapower
                            performs on every runpf (pandaPowerBackend.py:1086 _pf_init = "dc"; LightSimBackend-style warm start).
                            Same solution within the solver tolerance, fewer iterations: n_iter and the last digits differ
                            from the reference's.  Default 0 = the reference's algorithm. */
  int32_t track_cooldown; /* != 0: maintain the environment's LINE COOLDOWNS (BaseEnv._times_before_line_status_actionable =
                            obs.time_before_cooldown_line, Environment/baseEnv.py:3352-3358, 2590-2597) at every converged step: decremented,
                            set to nb_ts_reco for a line the protections trip in the step, raised to the remaining duration of a
                            maintenance / hazard under way (the uploaded outage tables).  0 (what a zero-initialised struct says, like every
                            other field): the counters are left alone (gpf_step: always).  Read with gpf_get_cooldown /
                            gpf_get_trajectory_cooldown; cleared by gpf_reset_lanes and by an auto-reset, copied by gpf_copy_lanes,
                            gpf_fanout_n1 and gpf_simulate_batch -- whose scratch step never maintains them (one look-ahead step on the
                            forecast tables: the source's counters are what obs.simulate starts from, _obsEnv.py:321-428).
                            (Cooldowns caused by the agents' own line / substation actions belong to the caller: a DoNothing step has none.) */
  int32_t nb_ts_reco;    /* Parameters.NB_TIMESTEP_RECONNECTION (default 10; >= 0): the cooldown of a line the protections trip;
                            only read when track_cooldown != 0 */
} gpf_step_opts;
/* n_steps consecutive DoNothing env.step (t0, t0+1, ...) of every lane in ONE launch.  Every step does the whole of gpf_step;
 * between the steps of a launch the lane state stays on chip and whatever only depends on the topology (element->bus maps, bus
 * types, Ybus, the factored DC matrix) is kept until a line trips or a lane fails.
 * n_steps = 1 (an agent that acts between any two steps: the reference's loop, Environment/baseEnv.py:3562-3931): a lane whose topology
 * row, shunt buses and shunt set-points are those the engine was created with loads that topology-only state from one blob the engine
 * keeps (written by the first such launch) instead of rebuilding it; every other lane rebuilds.  Results are bit-identical either way,
 * and equal to the same steps inside one multi-step launch bit for bit (GRIDPF_KEEP=0 at gpf_create: always rebuild).
 * What is retrievable afterwards: the getters (gpf_get_results, gpf_get_step_outputs, device views) return the LAST step only --
 * without a trajectory buffer each step overwrites the lane's result row, so the observations of the earlier steps never exist in
 * HBM.  gpf_set_trajectory(h, cap, GPF_TRAJ_OBS) keeps the complete backend observation of EVERY step (what BaseEnv.step hands
 * to the observation after each env.step: Environment/baseEnv.py:3562-3931 -> Observation/completeObservation.py:140-211 reads
 * the backend's flows / voltages / injections, topo_vect and line status); GPF_TRAJ_RHO keeps rho + status only.
 * n_steps must not exceed the capacity of a trajectory buffer that is set.  Asynchronous. */
int gpf_step_n(gpf_handle h, int32_t t0, int32_t n_steps, const gpf_step_opts* opts);
/* Per-lane additive generator set-point delta in MW, [n_lanes][n_gen] (NULL: none): the redispatch the environment adds to the
 * chronics' prod_p every step (actual_dispatch, Environment/baseEnv.py:2211-2470, 3650-3700). */
int gpf_set_lane_redispatch(gpf_handle h, const float* delta_mw);
/* The environment's redispatching automaton (BaseEnv._compute_dispatch_vect, Environment/baseEnv.py:2211-2470), batched: for
 * every lane the redispatch the agents ask for is projected on pmin / pmax / ramp limits under the zero-sum constraint
 * sum(x) = rhs (rhs = storage power - curtailment + detached MW, :2335-2340) and added to the actual dispatch.
 * gpf_set_gen_limits: the generator characteristics of prods_charac.csv ([n_gen] each), eps_poly as BaseEnv._epsilon_poly.
 * gpf_redispatch: rows of lanes lane0..lane0+n-1, [n][n_gen]: new_p (chronics set-points of the step), prev_p (set-points of
 * the previous step incl. dispatch, BaseEnv._gen_activeprod_t_redisp), actual / target dispatch, modified (generators touched
 * by an action this step); ok[n] = 0 where the reference raises ImpossibleRedispatching (the row is then returned unchanged);
 * actual_after [n][n_gen] float.  apply != 0 also stores the result as the lanes' redispatch delta (gpf_set_lane_redispatch)
 * for the following gpf_step.  Synchronous. */
int gpf_set_gen_limits(gpf_handle h, const double* pmin, const double* pmax, const double* ramp_up, const double* ramp_down,
                       const uint8_t* redispatchable, double eps_poly);
int gpf_redispatch(gpf_handle h, int32_t lane0, int32_t n, const double* new_p, const double* prev_p, const double* actual,
                   const double* target, const uint8_t* modified, const double* rhs, int32_t apply, uint8_t* ok, float* actual_after);
/* ---- injection dynamics of the environment inside the stepped batch (BASELINE configs[3]: storage + redispatch actions) ------
 * What BaseEnv.step does to the generator / storage set-points between the chronics and the backend, evaluated by gpf_step_n at
 * EVERY step of a launch: the storage state of charge with its efficiencies, Emin / Emax clamping and losses
 * (Environment/baseEnv.py:2829-2905 _compute_storage, :2777-2790 _withdraw_storage_losses), the accumulation of the agents'
 * redispatch into the target dispatch (:2101-2115), the _make_redisp gate (:2188-2209) and the ramp- / pmin- / pmax-limited
 * zero-sum projection (:2211-2470 _compute_dispatch_vect; exact solution of the separable QP, as gpf_redispatch), then
 * prod_p = chronics + actual dispatch (:3830 set_redispatch) and the storage power (:3831 set_storage).  A lane whose projection is
 * infeasible ends its episode (status GPF_ST_REDISPATCH; ImpossibleRedispatching :3227-3247).  Curtailment
 * (_aux_handle_curtailment_without_limit, :2956-2982): renewable generators are capped at limit * pmax, the change of the curtailed
 * total joins the right-hand side of the projection.  An ILLEGAL redispatch -- the accumulated target beyond pmax - pmin or below
 * pmin - pmax (_prepare_redisp :2140-2173) -- is cancelled as the reference cancels it: taken back out of the target, the storage
 * part of the step undone (:3189-3212).  Not modelled: detachment, generator up / down times, the dispatch of switched-off generators
 * (Parameters.ALLOW_DISPATCH_GEN_SWITCH_OFF = False), LIMIT_INFEASIBLE_CURTAILMENT_STORAGE_ACTION.
 *   gpf_set_storage_params : storage_Emax / Emin / loss / charging & discharging efficiency / initial charge [n_storage], the
 *                            step length and Parameters.ACTIVATE_STORAGE_LOSS.
 *   gpf_set_env_dynamics   : on != 0 switches the dynamics on (needs gpf_set_gen_limits, and gpf_set_storage_params on a grid
 *                            with storage units) and resets the state of every lane (dispatch 0, initial charge); tol_poly as
 *                            BaseEnv._tol_poly.  Needs n_gen and n_storage <= the lanes an instance owns (16 / 32 / 64).
 *   gpf_set_lane_actions   : the agents' actions of the NEXT launch: redispatch [n_lanes][n_gen] MW (added to the target dispatch
 *                            by the launch's first step, then consumed) and storage power [n_lanes][n_storage] MW (applied by the
 *                            first step only -- grid2op's semantics of a storage action -- or, hold_storage != 0, by every step
 *                            until replaced); NULL = none.  The arrays are copied into pinned staging before the call returns;
 *                            the upload rides the engine's stream, nothing is synchronised.
 *   gpf_set_gen_renewable  : gen_renewable [n_gen] (NULL: no curtailment);  gpf_set_lane_curtailment: the curtailment action of
 *                            the NEXT launch, [n_lanes][n_gen] ratios of pmax in [0, 1], -1 = no change (consumed by the first
 *                            step; the limits then live in the lanes' state, like BaseEnv._limit_curtailment).
 *   gpf_get/set_env_state  : target / actual dispatch, previous set-points (_gen_activeprod_t_redisp), already-modified mask
 *                            [n][n_gen], state of charge [n][n_storage], previous storage amount [n], curtailment limits
 *                            [n][n_gen], previous curtailed total [n] (what an environment restored from an observation hands
 *                            over, baseEnv.py:4879-4882); any pointer may be NULL.
 * gpf_reset_lanes also resets the dynamics of the lanes. */
#define GPF_ST_REDISPATCH 6   /* the redispatch projection is infeasible: game over (ImpossibleRedispatching) */
int gpf_set_storage_params(gpf_handle h, const double* emax, const double* emin, const double* loss, const double* eff_charge,
                           const double* eff_discharge, const float* charge0, double delta_time_seconds, int32_t activate_loss);
int gpf_set_env_dynamics(gpf_handle h, int32_t on, double tol_poly);
int gpf_set_lane_actions(gpf_handle h, const float* redispatch, const float* storage_power, int32_t hold_storage);
/* The same hand-over for agents that live ON THE DEVICE (a policy network next to the engine): the caller has written the actions of
 * the next launch into the engine's own action buffers -- gpf_device_pointers_n entries 22 (redispatch [lanes][n_gen] MW), 23 (storage
 * power [lanes][n_storage] MW), 24 (curtailment [lanes][n_gen], ratios in [0, 1] or -1: NOT validated here) -- on the engine's stream
 * or ordered before the next launch; the flags say which of them hold an action (the others count as "none").  Nothing crosses PCIe,
 * nothing is synchronised.  Consumption is as for gpf_set_lane_actions / gpf_set_lane_curtailment: the launch's first step takes the
 * actions; afterwards the redispatch buffer (and the storage buffer unless hold_storage) counts as empty until declared again. */
int gpf_lane_actions_on_device(gpf_handle h, int32_t redispatch, int32_t storage_power, int32_t curtailment, int32_t hold_storage);
int gpf_set_gen_renewable(gpf_handle h, const uint8_t* renewable);
int gpf_set_lane_curtailment(gpf_handle h, const float* limit);
int gpf_get_env_state(gpf_handle h, int32_t lane0, int32_t n, float* target, float* actual, float* prev_p, uint8_t* already_modified,
                      float* charge, float* amount_prev, float* curtail_limit, float* curtail_prev);
/* count[n]: steps since the lane's last reset whose action was CANCELLED as an illegal redispatch (what BaseEnv.step reports as
 * info["is_illegal_redisp"] / the IllegalRedispatching exception of that step, baseEnv.py:2140-2173, 3400-3425); copied with the lane
 * by gpf_copy_lanes / gpf_fanout_n1 / gpf_simulate_batch, cleared by gpf_reset_lanes and auto-reset. */
int gpf_get_env_illegal(gpf_handle h, int32_t lane0, int32_t n, int32_t* count);
/* overwrite the counters (a lane moved between engines / devices through the host carries its count: ShardedEngine.copy_lanes) */
int gpf_set_env_illegal(gpf_handle h, int32_t lane0, int32_t n, const int32_t* count);
int gpf_set_env_state(gpf_handle h, int32_t lane0, int32_t n, const float* target, const float* actual, const float* prev_p,
                      const uint8_t* already_modified, const float* charge, const float* amount_prev, const float* curtail_limit,
                      const float* curtail_prev);

/* ---- batched obs.simulate (Observation/baseObservation.py:3365-3670 simulate -> Environment/_obsEnv.py: the forecast
 * injections of `time_step` steps ahead + a candidate action on a copy of the observation's grid state, one env.step of that copy) --
 * gpf_upload_forecasts: the *_forecasted tables of the uploaded chronics (Chronics/gridStateFromFileWithForecasts.py:311-353:
 * forecast h_id of chronics row r is row n_horizons * r + h_id), [n_tables][T][n_horizons][n_chron] float, same row layout as
 * the chronics; NULL removes them.
 * gpf_simulate_batch: for each of the n_src source lanes (environments whose current observation is the step at time index t_obs)
 * and each of the n_act candidate actions, lane dst_lane0 + b * n_act + k becomes a copy of source lane b (topology, shunts,
 * storage / shunt set-points, redispatch delta, jitter, protection counters) with action k applied to the topology as
 * _BackendAction.__iadd__ applies it (Action/_backendAction.py:836-919: line status first -- a reconnected end goes back to its
 * last known busbar, `last_bus` [n_src][dim_topo] or NULL = busbar 1 --, then change_bus, then set_bus, then lines with one open
 * end are opened / lines reconnected by a bus assignment get their other end back), and ONE launch steps all n_src * n_act lanes
 * with the injections of the forecast `time_step` steps ahead (time_step = 0: the current chronics row, i.e. the observation's own
 * injections) under `opts` (cascade, thermal limits, is_dc ... as gpf_step_n; one step).  Results: the getters on the
 * destination range (gpf_get_results, gpf_get_step_outputs -> rho).  The destination lanes are scratch lanes: their chronics
 * cursor and counters are overwritten.  Actions: act_off[n_act + 1] offsets into act_items[][3] = {kind, id, value}:
 *   GPF_ACT_SET_BUS {topo_vect position, bus (-1 | 1..n_busbar)}   GPF_ACT_CHANGE_BUS {topo_vect position, -}
 *   GPF_ACT_SET_LINE_STATUS {line id, +1 | -1}                      GPF_ACT_CHANGE_LINE_STATUS {line id, -}
 *   GPF_ACT_SET_SHUNT_BUS {shunt id, bus}
 * With the injection dynamics on (gpf_set_env_dynamics) every scratch lane also inherits its source's dispatch / storage /
 * curtailment state (what _ObsEnv is initialised with) and takes ONE do-nothing step of the dynamics on the simulated injections
 * (the candidates are topology actions); the sources' state and the per-lane actions waiting for the next gpf_step_n are untouched.
 * A scratch lane cannot be the SOURCE of a later call -- its chronics cursor is an absolute row (of the forecast tables when it simulated
 * a forecast), not an offset to the time index (GPF_E_INVALID; gpf_set_lane_chronics puts every lane back on the chronics).
 * Asynchronous launch (the topology bookkeeping before it synchronises once). */
#define GPF_ACT_SET_BUS 0
#define GPF_ACT_SET_LINE_STATUS 1
#define GPF_ACT_CHANGE_BUS 2
#define GPF_ACT_CHANGE_LINE_STATUS 3
#define GPF_ACT_SET_SHUNT_BUS 4
int gpf_upload_forecasts(gpf_handle h, int32_t n_tables, int32_t T, int32_t n_horizons, const float* data);
int gpf_simulate_batch(gpf_handle h, int32_t t_obs, int32_t time_step, int32_t n_src, const int32_t* src_lanes, int32_t n_act,
                       const int32_t* act_off, const int32_t* act_items, const int32_t* last_bus, int32_t dst_lane0,
                       const gpf_step_opts* opts);

/* Trajectory buffers of multi-step launches (n_steps_cap = 0 or what = 0 releases them).
 *   GPF_TRAJ_RHO: rho [cap][n_lanes][n_line] and status [cap][n_lanes] of every step of the last gpf_step_n.
 *   GPF_TRAJ_OBS: in addition the complete backend observation of every step -- results row out [cap][n_lanes][n_out]
 *                 (layout of gpf_get_results), topo_vect [..][dim_topo], shunt_bus [..][n_shunt], line_status [..][n_line]:
 *                 every step of the launch then writes its rows to HBM (1 observation per env.step, as the reference
 *                 returns one per BaseEnv.step); the lane's own rows still return the last step.
 * gpf_get_trajectory / gpf_get_trajectory_obs copy steps [step0, step0+n_steps) of lanes [lane0, lane0+n) -- only steps written
 * by the LAST gpf_step_n are retrievable (GPF_E_INVALID beyond).  Any output pointer may be NULL. */
#define GPF_TRAJ_RHO 1
#define GPF_TRAJ_OBS 2
int gpf_set_trajectory(gpf_handle h, int32_t n_steps_cap, int32_t what);
int gpf_get_trajectory(gpf_handle h, int32_t step0, int32_t n_steps, int32_t lane0, int32_t n, float* rho, int8_t* status);
int gpf_get_trajectory_obs(gpf_handle h, int32_t step0, int32_t n_steps, int32_t lane0, int32_t n, float* out, int32_t* topo_vect,
                           int32_t* shunt_bus, uint8_t* line_status);
/* Episode bookkeeping of the batched steps: done [n] (1: the lane's last step ended its episode), steps_and_resets [n][2]
 * {steps survived since the last (auto-)reset, number of auto-resets}. */
int gpf_get_episode(gpf_handle h, int32_t lane0, int32_t n, uint8_t* done, int32_t* steps_and_resets);
/* The environment's protection counters of lanes lane0..lane0+n-1 (BaseEnv._timestep_overflow, Environment/baseEnv.py:3346-3370:
 * consecutive steps each line has spent above its thermal limit), [n][n_line]: what an environment restored from an observation
 * hands over (Environment/_obsEnv.py init copies obs.timestep_overflow).  gpf_step / gpf_step_n maintain them on the device. */
int gpf_set_overflow_count(gpf_handle h, int32_t lane0, int32_t n, const int32_t* overflow_count);
/* The lanes' line cooldowns (gpf_step_opts::track_cooldown), [n][n_line]; gpf_set_cooldown: what an environment restored from an observation
 * hands over (obs.time_before_cooldown_line).  gpf_get_trajectory_cooldown: the counters after every step of the last multi-step launch,
 * int16 [n_steps][n][n_line] (saturating at 32767), kept with any trajectory (gpf_set_trajectory). */
int gpf_get_cooldown(gpf_handle h, int32_t lane0, int32_t n, int32_t* line_cooldown);
int gpf_set_cooldown(gpf_handle h, int32_t lane0, int32_t n, const int32_t* line_cooldown);
int gpf_get_trajectory_cooldown(gpf_handle h, int32_t step0, int32_t n_steps, int32_t lane0, int32_t n, int16_t* line_cooldown);
/* rho = a_or / thermal_limit (backend.py:1145-1168) and overflow counters of the last gpf_step. */
int gpf_get_step_outputs(gpf_handle h, int32_t lane0, int32_t n, float* rho, int32_t* overflow_count,
                         int32_t* disc_round);

/* ---- DC sensitivity (PTDF) path ------------------------------------------------------------------------------
 * The reference solves B' theta = P from scratch on every DC power flow (pp.rundcpp, pandaPowerBackend.py:1090).
 * For a FIXED topology the DC branch flows are linear in the bus injections: p_or = PTDF * P_bus.
 * gpf_ptdf_build factorises the DC system of the topology currently held by `lane` (host side, once) and keeps the
 * PTDF on the device; gpf_ptdf_flows evaluates lanes [lane0, lane0+n) from their current injection rows (as left
 * by gpf_set_injections / gpf_step) with one FP64 MFMA GEMM (asynchronous); gpf_get_ptdf_flows copies the active
 * power flows at the origin side (MW, float32 [n][n_line]; the extremity side is the negative, DC has no losses).
 * Errors: no in-service slack generator / islanded topology -> GPF_E_INVALID. */
int gpf_ptdf_build(gpf_handle h, int32_t lane);
int gpf_ptdf_get(gpf_handle h, double* ptdf /* [n_line][n_sub*n_busbar] row-major, MW per MW */);
int gpf_ptdf_flows(gpf_handle h, int32_t lane0, int32_t n);
int gpf_get_ptdf_flows(gpf_handle h, int32_t lane0, int32_t n, float* p_or);
/* ---- the same for PER-LANE topologies: the DC matrices of every DISTINCT topology of a lane range factorised ON THE DEVICE ------------
 * The reference factorises B' of whatever topology each environment has on every DC call (pp.rundcpp, pandaPowerBackend.py:1090;
 * N1Reward once per contingency, grid2op/Reward/n1Reward.py:70-99).  gpf_ptdf_build_batch groups lanes [lane0, lane0 + n) by their
 * CURRENT topology rows (as on the device: lines tripped by a cascade included) into classes and, in ONE launch with one workgroup per
 * class, assembles the reduced B', inverts it with a blocked Gauss-Jordan whose panel / trailing updates run on the FP64 matrix cores
 * (v_mfma_f64_16x16x4_f64; 2 n^3 flops per class) and forms PTDF^T and (with_lodf != 0) the LODF table of the class
 * (grid2op_amd/csrc/gridpf_ptdf_batch.hpp).  The host only does the integer work per class (live buses, compact numbering, the
 * connectivity check of rundcpp(check_connectivity=True)).  Afterwards gpf_ptdf_flows / gpf_ptdf_flows_rows / gpf_lodf_screen evaluate
 * every lane against the tables of ITS class (gpf_ptdf_flows and gpf_lodf_screen must be called on exactly [lane0, lane0 + n),
 * gpf_ptdf_flows_rows needs the batch built for all lanes); lanes of a class without sensitivities (islanded topology, no in-service
 * slack, singular pivot) get NaN flows instead of an error.  gpf_ptdf_build switches back to the single-topology tables.
 * Capacity: at most 256 active non-reference buses per topology (GPF_E_CAPACITY).  *n_classes (may be NULL) = distinct topologies. */
int gpf_ptdf_build_batch(gpf_handle h, int32_t lane0, int32_t n, int32_t with_lodf, int32_t* n_classes);
/* lane_class[n]: class of every lane of the built range; class_status[n_classes]: 0 ok, 1 singular pivot, 2 islanded, 3 no in-service
 * slack; class_n[n_classes]: dimension of the reduced B' of the class; *kernel_ms: duration of the build kernel (HIP events).  Any
 * pointer may be NULL. */
int gpf_ptdf_batch_info(gpf_handle h, int32_t* lane_class, int32_t* class_status, int32_t* class_n, double* kernel_ms);
/* tables of one class in the layout of gpf_ptdf_get: ptdf [n_line][n_sub*n_busbar], lodf [n_line][n_line] (either may be NULL) */
int gpf_ptdf_batch_get(gpf_handle h, int32_t cls, double* ptdf, double* lodf);
/* The same over n_rows CONSECUTIVE CHRONICS ROWS of every lane in ONE launch (M = n_lanes x n_rows rows of the GEMM): row j of lane k
 * is chronics row (t0 + j + lane_offset[k]) mod T of table lane_table[k] turned into injections exactly as gpf_step does (loads x
 * lane_scale, non-slack prod_p rescaled to rebalance x sum(load) when rebalance > 0, + the lane's redispatch delta; storage and shunt
 * set-points from the lane's injection row), i.e. the DC flows of the next n_rows DoNothing env steps of the whole batch for the
 * fixed topology of gpf_ptdf_build -- what rundcpp would return at each of them (pandaPowerBackend.py:1090).  Asynchronous;
 * gpf_get_ptdf_flows_rows copies rows [row0, row0 + n_rows) of lanes [lane0, lane0 + n): float32 [n_rows][n][n_line] MW at the origin. */
int gpf_ptdf_flows_rows(gpf_handle h, int32_t t0, int32_t n_rows, double rebalance);
int gpf_get_ptdf_flows_rows(gpf_handle h, int32_t row0, int32_t n_rows, int32_t lane0, int32_t n, float* p_or);
/* DC N-1 screening on top of the PTDF path (what N1Reward / obs.simulate loops do one contingency at a time,
 * grid2op/Reward/n1Reward.py:70-99): post-outage flows are f_l + LODF[l][k] * f_k, so for every lane of the range and
 * every single-line outage k this returns the largest post-outage loading max_l |f_l + LODF[l][k] f_k| / cap_mw[l]
 * (cap_mw NULL: the largest |flow| in MW), computed from the flows left by the last gpf_ptdf_flows; outages that island
 * the grid give +inf.  Synchronous; worst is [n][n_line]. */
int gpf_lodf_screen(gpf_handle h, int32_t lane0, int32_t n, const float* cap_mw, float* worst);

/* ---- measurement ---------------------------------------------------------------------------------------- */
int gpf_sync(gpf_handle h);
/* Event timing of the solver launches on the handle's stream.  mode 0: off.  mode 1 (window): ONE event pair around all
 * launches issued until the next gpf_get_kernel_time / gpf_set_profiling call -- no per-launch events, so back-to-back
 * launches stay back-to-back (a per-launch pair costs ~7 us of stream time per launch on MI355X); the window time divided
 * by the launch count is the average launch duration when the stream never runs dry.  mode 2: every launch is bracketed
 * by its own event pair (exact per-kernel durations, perturbs throughput).  mode 3 (inside a window): the window ENDS at this
 * point of the stream -- the closing event is recorded right behind the launches issued so far (asynchronous, no wait), so that
 * host work between the last launch and the next gpf_get_kernel_time (a barrier with other ranks, say) is not part of the window;
 * without it the window ends when gpf_get_kernel_time / gpf_set_profiling is called. */
int gpf_set_profiling(gpf_handle h, int32_t mode);
/* Sum of the event-measured durations (ms) and number of solver launches since the last call (closes the running
 * window of mode 1 and opens the next one). */
int gpf_get_kernel_time(gpf_handle h, double* total_ms, int64_t* n_launches);
/* out[2] = {gpf_step / gpf_step_n / gpf_simulate_batch step launches since gpf_create, kernel dispatches they issued}.  A batch of a few
 * residency rounds of equal lanes (e.g. 2 048 lanes of a 118-substation grid: 2 x the 1 024 resident blocks) goes out as one dispatch per
 * round, back to back on the engine's stream (gridpf_launch_step.hip); everything else is one dispatch per launch. */
int gpf_get_counters(gpf_handle h, int64_t out[2]);
/* Diagnostics (no reference counterpart): the kernel configuration a launch over ALL lanes would use right now.
 * out[0] busbars per block (1: single-busbar kernel; 2, 3: NB = n_busbar kernel), out[1] instances per wavefront,
 * out[2] wavefronts per instance, out[3] static-table staging tier (0 global memory, 1 program + pair table in LDS, 2 all),
 * out[4] Ybus blocks in registers, out[5] LDS layout keeps the factored DC matrix, out[6] dynamic LDS bytes per block,
 * out[7] topology-class launch. */
int gpf_get_plan(gpf_handle h, int32_t out[8]);
/* Raw device pointers + the stream, for zero-copy interop (grid2op_amd/engine.py: PowerFlowEngine.device_views wraps them as
 * torch tensors).  ptrs[0..21] = inj, topo, shunt_bus, out, topo_vect, line_status, status, chronics, rho, overflow_count, done,
 * episode, bus_vm, bus_va, shunt_bus_out, disc_round, then the trajectory buffers (NULL when not set): traj_rho, traj_status,
 * traj_out, traj_topo_vect, traj_shunt_bus, traj_line_status (rows are padded to gpf_lane_capacity lanes; trajectory buffers are
 * [cap][gpf_lane_capacity][row]); 22..27 = the environment dynamics (NULL while they are off): the action buffers redispatch
 * [lanes][n_gen], storage power [lanes][n_storage], curtailment [lanes][n_gen] (gpf_lane_actions_on_device), then target dispatch,
 * actual dispatch [lanes][n_gen] and state of charge [lanes][n_storage] (obs.target_dispatch / actual_dispatch / storage_charge);
 * stream = hipStream_t */
#define GPF_N_DEVICE_POINTERS 28
int gpf_device_pointers(gpf_handle h, void** ptrs /* [GPF_N_DEVICE_POINTERS] */, void** stream);
/* The same with the length of the caller's array: entries beyond n_ptrs are not written, entries beyond the library's count are
 * NULL -- a caller built against an older / newer header cannot be overrun. */
int gpf_device_pointers_n(gpf_handle h, void** ptrs, int32_t n_ptrs, void** stream);

/* ---- grid-specialised step kernels (run-time compilation; grid2op_amd/csrc/gridpf_jit.hip) -------------------------------------
 * The shipped (ahead-of-time) kernels serve every grid: sizes, offsets of the static tables / result rows and the header of the
 * symbolic program reach them through a parameter block.  gpf_jit_enable() switches the engine's solver launches (gpf_step,
 * gpf_step_n, gpf_simulate_batch, gpf_runpf, gpf_solve_lane) to kernels compiled for THIS grid, in which those numbers are literals: at the first launch of
 * each kernel variant the library writes a header with the grid's numbers, compiles the unchanged kernel source of
 * <src_dir>/gridpf_sparse.hpp for that variant (hipcc --genco, about a second), loads the code object and launches it from then on;
 * code objects are cached in <cache_dir> by a hash of header + variant + sources, so a grid is compiled once per machine.
 * Results are BIT-IDENTICAL to the ahead-of-time kernels (same source, same arithmetic in the same order).  Nothing else changes: same
 * buffers, same calls.  The one-power-flow-per-lane kernels of gpf_runpf / gpf_solve_lane are specialised the same way.
 *   src_dir   directory with the kernel sources (NULL: "csrc" next to the library)
 *   cache_dir NULL: $GRIDPF_JIT_CACHE, else "_jit_cache" next to the library; when that is not writable or not private (owned by
 *             another user / group- or world-writable / a symbolic link): $XDG_CACHE_HOME/gridpf_jit, $HOME/.cache/gridpf_jit -- created 0700
 *             and held to the same rule.  Code objects are loaded without an integrity check: only private directories are trusted.
 * Code objects built ahead of time (grid2op_amd/_aot, made by __graft_entry__.build() for the grids of grid2op_amd/aot/) are used first
 * and need no compiler at run time.  The compiler is $GRIDPF_HIPCC, else /opt/rocm/bin/hipcc, else hipcc on PATH, started without a
 * shell; GPF_E_UNSUPPORTED when neither a compiler + cache nor ahead-of-time objects exist, when the sources are absent or the device
 * is not gfx950 -- the engine then simply keeps the ahead-of-time kernels.  A variant that fails to compile / load is reported on stderr and
 * in gpf_jit_info and runs ahead-of-time.  GRIDPF_JIT=1 in the environment enables it at gpf_create. */
int gpf_jit_enable(gpf_handle h, const char* src_dir, const char* cache_dir);
int gpf_jit_disable(gpf_handle h);
/* counts[6] = {enabled, variants compiled, variants loaded from the cache, variants failed, launches through specialised kernels,
 * variants loaded from the ahead-of-time directory ("_aot" next to the library)};
 * seconds = time spent compiling / loading; text = the variants loaded so far (+ the last error).  Any pointer may be NULL. */
int gpf_jit_info(gpf_handle h, int64_t* counts, double* seconds, char* text, size_t cap);
/* the header the specialised kernels are compiled with (one grid's numbers as C literals); *need = bytes incl. the terminator */
int gpf_jit_source(gpf_handle h, char* text, size_t cap, size_t* need);

#ifdef __cplusplus
}
#endif
#endif /* GRIDPF_H */
