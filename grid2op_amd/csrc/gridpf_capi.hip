// gridpf_capi.hip -- host side of libgridpf.so: the C ABI declared in include/gridpf.h.
// Owns the device buffers of one engine (static grid tables + lane-major per-lane state), one HIP
// stream, and launches the kernels of gridpf_sparse.hpp / gridpf_ptdf.hpp.  gfx950 only; no fallback path: if HIP is
// unavailable every entry point fails with GPF_E_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gridpf.h"
#include "gridpf_common.hpp"
#include "gridpf_sparse.hpp"
#include "gridpf_host.hpp"
#include "gridpf_ptdf.hpp"
#include "gridpf_ptdf_batch.hpp"
#include "gridpf_ptdf_group.hpp"
#include "gridpf_redispatch.hpp"
#include <string>
#include <thread>
#include <condition_variable>
#include <functional>
#include <unordered_map>
#include "gridpf_symbolic.hpp"

namespace gpf {
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
  int* d0 = const_cast<int*>(b.topo0) + (size_t)dst * g.dim_topo;
  for (int i = tid; i < g.dim_topo; i += blockDim.x) { const int v = (i == po || i == pe) ? -1 : st[i]; dt[i] = v; d0[i] = v; }
  const int* ss = b.shunt_bus + (size_t)src * g.n_shunt;
  int* ds = b.shunt_bus + (size_t)dst * g.n_shunt;
  for (int i = tid; i < g.n_shunt; i += blockDim.x) ds[i] = ss[i];
  // the contingency lane is the source's environment with one line out: its protection counters and line cooldowns are the source's
  // (as gpf_copy_lanes and simulate_prepare_kernel copy them; a step that tracks cooldowns would otherwise start from stale ones)
  if (b.overflow_count) for (int i = tid; i < g.n_line; i += blockDim.x) b.overflow_count[(size_t)dst * g.n_line + i] = b.overflow_count[(size_t)src * g.n_line + i];
  if (b.cooldown) for (int i = tid; i < g.n_line; i += blockDim.x) b.cooldown[(size_t)dst * g.n_line + i] = b.cooldown[(size_t)src * g.n_line + i];
}

// gpf_solve_lane: the lane's inputs straight from the pinned host block (device-mapped: one PCIe read per element, all in flight
// together) into the lane's rows, and its result rows straight back into it -- two dispatches instead of eleven staged copies
// (5.6 us each on the stream: 60 of the 73 us a one-lane runpf took).  Offsets in bytes, 16-byte aligned pieces.
struct LaneBlob { size_t o_inj, o_topo, o_sb, o_out, o_tv, o_sbo, o_ls, o_st, o_vm, o_va; };
__global__ void lane_scatter_kernel(GridDev g, Bufs b, int lane, const unsigned char* __restrict__ blob, LaneBlob o) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* inj = reinterpret_cast<const double*>(blob + o.o_inj);
  const int* topo = reinterpret_cast<const int*>(blob + o.o_topo);
  const int* sb = reinterpret_cast<const int*>(blob + o.o_sb);
  double* dinj = b.inj + (size_t)lane * g.n_inj;
  int* dt = b.topo + (size_t)lane * g.dim_topo;
  int* d0 = const_cast<int*>(b.topo0) + (size_t)lane * g.dim_topo;
  int* ds = b.shunt_bus + (size_t)lane * g.n_shunt;
  for (int i = tid; i < g.n_inj; i += nt) dinj[i] = inj[i];
  for (int i = tid; i < g.dim_topo; i += nt) { const int v = topo[i]; dt[i] = v; d0[i] = v; }
  for (int i = tid; i < g.n_shunt; i += nt) ds[i] = sb[i];
}
__global__ void lane_gather_kernel(GridDev g, Bufs b, int lane, unsigned char* __restrict__ blob, LaneBlob o) {
  const int tid = threadIdx.x, nt = blockDim.x;
  float* out = reinterpret_cast<float*>(blob + o.o_out);
  int* tv = reinterpret_cast<int*>(blob + o.o_tv);
  int* sbo = reinterpret_cast<int*>(blob + o.o_sbo);
  unsigned char* ls = blob + o.o_ls;
  int* st = reinterpret_cast<int*>(blob + o.o_st);
  double* vm = reinterpret_cast<double*>(blob + o.o_vm);
  double* va = reinterpret_cast<double*>(blob + o.o_va);
  for (int i = tid; i < g.n_out; i += nt) out[i] = b.out[(size_t)lane * g.n_out + i];
  for (int i = tid; i < g.dim_topo; i += nt) tv[i] = b.topo_out[(size_t)lane * g.dim_topo + i];
  for (int i = tid; i < g.n_shunt; i += nt) sbo[i] = b.shunt_bus_out[(size_t)lane * g.n_shunt + i];
  for (int i = tid; i < g.n_line; i += nt) ls[i] = b.line_status[(size_t)lane * g.n_line + i];
  if (tid < 4) st[tid] = b.status[(size_t)lane * 4 + tid];
  for (int i = tid; i < g.nb_tot; i += nt) { vm[i] = b.bus_vm[(size_t)lane * g.nb_tot + i]; va[i] = b.bus_va[(size_t)lane * g.nb_tot + i]; }
  __threadfence_system();
}

// gpf_simulate_batch: topology / shunt rows of the source lanes -> one dense staging buffer (a single device-to-host copy)
__global__ void gather_topo_kernel(GridDev g, Bufs b, const int* __restrict__ src_lanes, int n_src, int* __restrict__ dst) {
  const int k = blockIdx.x;
  if (k >= n_src) return;
  const int src = src_lanes[k];
  const int w = g.dim_topo + g.n_shunt;
  for (int i = threadIdx.x; i < g.dim_topo; i += blockDim.x) dst[(size_t)k * w + i] = b.topo[(size_t)src * g.dim_topo + i];
  for (int i = threadIdx.x; i < g.n_shunt; i += blockDim.x) dst[(size_t)k * w + g.dim_topo + i] = b.shunt_bus[(size_t)src * g.n_shunt + i];
}

// gpf_simulate_batch: destination lane d = dst0 + b * n_act + k takes over everything of source lane b that is not topology
// (injection row, protection counters, chronics table / jitter / redispatch delta) and gets the chronics-table row to step on:
// the source's current row idx = (t_obs + offset) mod T for time_step 0, else forecast row n_h * idx + time_step - 1.
__global__ void simulate_prepare_kernel(GridDev g, Bufs b, const int* __restrict__ src_lanes, int n_act, int n_dst, int dst0, int t_obs,
                                        int T, int n_h, int time_step, int* __restrict__ lane_table, int* __restrict__ lane_offset,
                                        float* __restrict__ lane_scale, float* __restrict__ lane_gen_delta) {
  const int q = blockIdx.x;
  if (q >= n_dst) return;
  const int src = src_lanes[q / n_act], dst = dst0 + q, tid = threadIdx.x;
  for (int i = tid; i < g.n_inj; i += blockDim.x) b.inj[(size_t)dst * g.n_inj + i] = b.inj[(size_t)src * g.n_inj + i];
  for (int i = tid; i < g.n_line; i += blockDim.x) b.overflow_count[(size_t)dst * g.n_line + i] = b.overflow_count[(size_t)src * g.n_line + i];
  for (int i = tid; i < g.n_line; i += blockDim.x) b.cooldown[(size_t)dst * g.n_line + i] = b.cooldown[(size_t)src * g.n_line + i];
  if (lane_scale) for (int i = tid; i < 2 * g.n_load; i += blockDim.x) lane_scale[(size_t)dst * 2 * g.n_load + i] = lane_scale[(size_t)src * 2 * g.n_load + i];
  if (lane_gen_delta) for (int i = tid; i < g.n_gen; i += blockDim.x) lane_gen_delta[(size_t)dst * g.n_gen + i] = lane_gen_delta[(size_t)src * g.n_gen + i];
  if (tid == 0) {
    lane_table[dst] = lane_table[src];
    int idx = (t_obs + lane_offset[src]) % T;
    if (idx < 0) idx += T;
    lane_offset[dst] = time_step == 0 ? idx : n_h * idx + (time_step - 1);
    b.done[dst] = 0;
  }
}

// the environment-dynamics state of the source lane -> the scratch lane (simulate: _ObsEnv starts from the environment's own
// _target_dispatch / _actual_dispatch / _gen_activeprod_t_redisp / storage charge / curtailment limits, Environment/_obsEnv.py)
__global__ void simulate_env_copy_kernel(EnvDyn E, int n_gen, int n_sto, const int* __restrict__ src_lanes, int n_act, int n_dst, int dst0) {
  const int q = blockIdx.x;
  if (q >= n_dst) return;
  const int src = src_lanes[q / n_act], dst = dst0 + q, tid = threadIdx.x;
  for (int i = tid; i < n_gen; i += blockDim.x) {
    const size_t s_ = (size_t)src * n_gen + i, d_ = (size_t)dst * n_gen + i;
    E.target[d_] = E.target[s_]; E.actual[d_] = E.actual[s_]; E.prev_p[d_] = E.prev_p[s_]; E.already[d_] = E.already[s_]; E.limit[d_] = E.limit[s_];
  }
  for (int i = tid; i < n_sto; i += blockDim.x) E.charge[(size_t)dst * n_sto + i] = E.charge[(size_t)src * n_sto + i];
  if (tid == 0) { E.amount_prev[dst] = E.amount_prev[src]; E.curt_prev[dst] = E.curt_prev[src]; E.fresh[dst] = E.fresh[src]; E.illegal[dst] = E.illegal[src]; }
}

}  // namespace gpf

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(GPF_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));         \
  } while (0)

// gpf_create(device = GPF_DEVICE_NONE): a HEADER-ONLY handle is being built -- every host-side step of gpf_create runs (symbolic analysis, static
// tables, launch planning), nothing is allocated on or copied to a device (there may be none).  Thread-local: set for the duration of that call.
static thread_local bool g_dry_create = false;

template <typename T>
struct DevArr {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0 || g_dry_create) { p = nullptr; return hipSuccess; }
    return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
  }
  hipError_t upload(const T* src, size_t count) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0 || g_dry_create) return e;
    return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    cap = 0;
  }
  size_t cap = 0;                          // elements allocated (ensure / put: grow-only buffers reused across calls)
  hipError_t ensure(size_t count) {        // room for `count` elements; contents undefined afterwards when it had to grow
    if (count <= cap && p) { n = count; return hipSuccess; }
    if (p) (void)hipFree(p);
    p = nullptr;
    const size_t want = count + count / 4;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(want, 1) * sizeof(T));
    cap = e == hipSuccess ? std::max<size_t>(want, 1) : 0;
    n = e == hipSuccess ? count : 0;
    return e;
  }
  hipError_t put(const T* src, size_t count, hipStream_t stream) {   // ensure + asynchronous upload on `stream` (src must stay alive until it ran)
    hipError_t e = ensure(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, stream);
  }
};

}  // namespace

struct gpf_engine {
  int device = 0;
  bool dry = false;                     // header-only handle (gpf_create with GPF_DEVICE_NONE): no device resources, no launches
  int n_lanes = 0;
  hipStream_t stream = nullptr;
  gpf::GridDev g{};
  gpf::OutOff oo{};
  gpf_layout layout{};
  // static tables (device)
  DevArr<double> sub_vn_kv, br_y, br_bdc, gen_min_q, gen_max_q, shunt_fact;
  DevArr<int> line_or_sub, line_ex_sub, line_or_pos, line_ex_pos, gen_sub, gen_pos, load_sub, load_pos, sto_sub, sto_pos,
      shunt_sub;
  DevArr<unsigned char> gen_slack;
  // host copies needed to size launches
  std::vector<int> h_line_or_sub, h_line_ex_sub, h_line_or_pos, h_line_ex_pos, h_gen_sub, h_gen_pos, h_load_sub, h_load_pos,
      h_sto_sub, h_sto_pos, h_shunt_sub;
  std::vector<unsigned char> h_gen_slack;
  std::vector<double> h_init_inj;
  std::vector<int> h_init_topo, h_init_shunt_bus;
  // per-lane state (device)
  DevArr<double> inj, bus_vm, bus_va, work;
  DevArr<int> topo, shunt_bus, topo_out, shunt_bus_out, status, overflow_count, disc_round, lane_table, lane_offset, tmp_lines;
  DevArr<int> cooldown;                 // [B][n_line] line cooldowns of the environment (gpf::Bufs::cooldown)
  // topology-derived state of the reference topology shared by the lanes of one-step launches (gpf::KeepArgs): two blobs (Ybus in LDS /
  // in registers), allocated and keyed by the first gpf_step_n with n_steps = 1; GRIDPF_KEEP=0 at gpf_create turns it off
  DevArr<unsigned char> keep;
  gpf::KeepArgs keep_args{};
  int keep_launch = 0;
  bool keep_enabled = true;
  DevArr<unsigned short> maint_dur;     // [chron_tables][chron_T][n_line] remaining duration of the maintenance / hazard under way, or empty
  DevArr<short> traj_cool;              // [traj_cap][cap_lanes][n_line]
  DevArr<float> out, chron, lane_scale, thermal_limit, rho;
  DevArr<unsigned char> line_status, done;
  DevArr<int> topo0, episode;           // topology last sent by the host (auto-reset target); {steps survived, resets} per lane
  DevArr<float> lane_gen_delta, traj_rho;
  DevArr<unsigned char> maint;          // [chron_tables][chron_T][n_line] scheduled maintenance OR hazards (forced outages), or empty
  std::vector<unsigned char> h_maint, h_hazard;   // host copies of the two tables (the device holds their union)
  std::vector<unsigned short> h_outage_dur;       // gpf_upload_outage_durations: remaining durations given by the caller (else derived from the tables)
  std::vector<int> h_lane_table, h_lane_offset;   // host mirror of lane_table / lane_offset (gpf_simulate_batch: maintenance ahead of a source lane)
  std::vector<char> h_lane_forecast;              // 1: the lane is a scratch lane of gpf_simulate_batch (its offset is an absolute row, of the forecast tables for time_step > 0)
  // injection dynamics of the environment (gpf::EnvDyn)
  bool env_on = false, env_hold = false, env_act_r = false, env_act_s = false, sto_ready = false;
  int env_loss_on = 1;
  double env_coeff = 300.0 / 3600.0, env_tol = 1e-2;
  DevArr<float> env_target, env_actual, env_prev, env_charge, env_amount_prev, env_act_redisp, env_act_storage, sto_charge0;
  DevArr<float> env_limit, env_curt_prev, env_act_curtail;
  DevArr<int> env_illegal;              // [B] cancelled (illegal) actions since the reset
  DevArr<unsigned char> env_already, env_fresh, env_renewable;
  bool env_act_c = false, env_has_ren = false;
  DevArr<double> sto_emax, sto_emin, sto_loss, sto_effc, sto_effd;
  std::vector<float> h_charge0;
  DevArr<float> forecast;               // [chron_tables][chron_T][fc_h][n_chron] *_forecasted tables (gpf_upload_forecasts), or empty
  int fc_h = 0;
  DevArr<int> sim_src, sim_rows;        // gpf_simulate_batch staging: source lane list, gathered topology rows
  struct PtdfbCached { std::vector<int> row, desc, c2b; };      // gpf_ptdf_build_batch: descriptor of a topology row seen before
  std::unordered_map<uint64_t, std::vector<PtdfbCached>> ptdfb_cache;
  size_t ptdfb_cache_n = 0;
  int ptdfb_cache_stride = 0;
  unsigned char* res_pin = nullptr;     // pinned block of gpf_get_results_pinned
  size_t res_pin_bytes = 0;
  float* act_pin = nullptr;             // pinned staging of gpf_set_lane_actions / gpf_set_lane_curtailment (redispatch | storage | curtailment)
  size_t act_pin_n = 0;
  hipEvent_t act_up = nullptr;          // recorded behind the uploads that read it: the next call waits for it before rewriting the block
  int* sim_pin = nullptr;               // its pinned host block (grow-only): gathered source rows | candidate topology rows | candidate shunt rows
  size_t sim_pin_n = 0;
  DevArr<signed char> traj_status;
  DevArr<float> traj_out;               // per-step observation trajectory (GPF_TRAJ_OBS): [traj_cap][cap_lanes][n_out] ...
  DevArr<int> traj_topo, traj_shb;
  DevArr<unsigned char> traj_lstat;
  int traj_cap = 0;
  int traj_what = 0;                    // GPF_TRAJ_* bits of the current buffers
  int traj_valid = 0;                   // steps of the trajectory written by the last gpf_step_n
  bool has_delta = false;
  DevArr<double> rd_pmin, rd_pmax, rd_ru, rd_rd, rd_in;      // generator limits + staging of gpf_redispatch
  DevArr<unsigned char> rd_redisp, rd_u8;
  DevArr<float> rd_after;
  double rd_eps = 1e-4;
  bool rd_ready = false;
  unsigned char* pin = nullptr;         // pinned host block of gpf_solve_lane (one lane in, one lane out), mapped into the device
  unsigned char* pin_dev = nullptr;     // its device-side address (hipHostGetDevicePointer)
  size_t pin_bytes = 0;
  GpfJit jit;                           // grid-specialised step kernels (gpf_jit_enable; gridpf_jit.hip)
  gpf::GridDev jit_g;                   // the grid-level part of the parameter block the specialisation was generated from
  gpf::OutOff jit_oo;
  gpf::SymDev jit_sym;
  int dcf = 0;                          // the NB == 1 LDS layout has room for the factored DC matrix (decided once at gpf_create)
  DevArr<double> d_init_inj;
  DevArr<int> d_init_topo, d_init_shunt_bus;
  int chron_T = 0, chron_tables = 0;
  bool has_scale = false;
  // per-lane capacity bookkeeping (host): number of active buses / NR unknowns of each lane
  std::vector<int> lane_nb, lane_nj;
  int init_nb = 0, init_nj = 0;
  // block-sparse path (kernel S)
  gpf::Symbolic sym;
  // DC sensitivity path (gridpf_ptdf.hpp)
  DevArr<int> ptdf_inj_bus;
  DevArr<double> ptdf_inj_w, ptdf_t;
  DevArr<float> ptdf_flow, lodf_worst, lodf_inv_cap;
  DevArr<float> ptdf_flow_rows;    // [rows][cap_lanes][line_pad] flows of the last gpf_ptdf_flows_rows
  int ptdf_rows_valid = 0;
  DevArr<float> lodf;              // [n_line][line_pad] line outage distribution factors of the PTDF topology, float32 (NaN column: islanding outage)
  std::vector<double> h_ptdf;      // [n_line][nb_tot]
  std::vector<double> h_br_bdc, h_shunt_fact;
  std::vector<int> h_gen_cnt;
  int ptdf_nb_pad = 0, ptdf_line_pad = 0;
  bool ptdf_ready = false;
  // per-lane topologies (gpf_ptdf_build_batch, gridpf_ptdf_batch.hpp): one PTDF^T / LODF block per distinct topology class of a lane range
  long long n_step_calls = 0, n_step_dispatches = 0;   // gpf_step_n calls / kernel dispatches they issued (gpf_get_counters)
  bool ptdf_batch = false;                 // the flows / screening calls run on the class tables of the last gpf_ptdf_build_batch
  int ptdfb_lane0 = 0, ptdfb_n = 0, ptdfb_classes = 0, ptdfb_slots = 0, ptdfb_kpad = 0, ptdfb_npad_max = 0, ptdfb_desc_stride = 0;
  DevArr<int> ptdfb_desc, ptdfb_order, ptdfb_blk_class, ptdfb_status;
  DevArr<double> ptdfb_work, ptdfb_t, ptdfb_inj_w;
  DevArr<float> ptdfb_lodf;
  std::vector<int> h_ptdfb_lane_class, h_ptdfb_status, h_ptdfb_desc;
  std::vector<std::vector<int>> h_ptdfb_bus;   // per class: compact bus index -> bus id (sub + (local - 1) * n_sub)
  double ptdfb_kernel_ms = 0.0;            // duration of the last build kernel (HIP events)
  // the build call returns once its kernel is QUEUED: class status + kernel time are fetched when somebody asks (gpf_ptdf_batch_info)
  hipEvent_t ptdfb_ev_a = nullptr, ptdfb_ev_b = nullptr;
  int* ptdfb_status_pin = nullptr; size_t ptdfb_status_pin_n = 0;
  bool ptdfb_pending = false;
  // device-side grouping + descriptors (gridpf_ptdf_group.hpp): their outputs, and the host mirrors fetched on demand
  DevArr<unsigned long long> ptdfg_hash;
  DevArr<int> ptdfg_lane_class, ptdfg_first, ptdfg_c2b, ptdfg_info;
  int* ptdfg_info_pin = nullptr;
  int* ptdfg_back_pin = nullptr; size_t ptdfg_back_pin_n = 0;     // lane -> class map + descriptor headers, queued behind the factorisation (pinned)
  bool ptdfb_prefetched = false;
  bool ptdfb_host_stale = false;           // h_ptdfb_lane_class / h_ptdfb_hdr are not what the device holds: ptdfb_fetch_host
  bool ptdfb_bus_stale = false;            // ... nor h_ptdfb_bus (only gpf_ptdf_batch_get reads it)
  std::vector<int> h_ptdfb_hdr;            // device path: the 4-int headers of the class descriptors (nr, n_act, n_pad, status); empty: h_ptdfb_desc has them
  DevArr<double> dc_inv_g;     // static DC inverse of the larger grids (gpf::SymDev::dc_inv_g)
  DevArr<double> stat_dbl;     // static blob of kernel S (gpf::StatOff)
  DevArr<int> stat_int;
  DevArr<int> flat_prog;       // flat programs of the substation graph (4 group widths)
  gpf::SymDev sym_dev{};
  gpf::DevParamsS h_params_s{};
  gpf::DevParamsS* d_params_s = nullptr;
  bool params_s_valid = false;
  // mixed batches: lanes without / with split substations are launched separately (single-busbar kernel / NB = n_busbar)
  DevArr<int> list_a, list_b, list_c;   // list_c: topology class of every lane of list_b
  // topology classes (gpf::TopoClassDev): bus-level graphs of the split topologies seen so far, each with its own symbolic program
  struct TopoClassHost { DevArr<int> tables, flat; gpf::TopoClassDev dev; int n_nodes, nslot, nslot_y; };
  std::vector<TopoClassHost*> classes;
  std::unordered_map<std::string, int> class_of_key;
  // host mirror of the topology last SENT for every lane (gpf_set_topology skips the per-lane bookkeeping when a lane is
  // re-sent unchanged: agents resend whole batches with few changes); first entry INT_MIN = unknown
  std::vector<int> h_lane_topo, h_lane_sb;
  bool dev_topo_dirty = false;           // a kernel may have rewritten topology rows (cascade trips, scheduled outages): the host mirrors are not the device rows any more
  std::vector<int> lane_class;          // per lane: topology class (-1: no split substation / classes disabled)
  DevArr<gpf::TopoClassDev> d_classes;  // device copy of classes[*].dev
  size_t d_classes_count = 0;
  bool no_classes = false;              // GRIDPF_NO_CLASSES=1: split lanes run the NB = n_busbar kernel
  bool no_partition = false;     // GRIDPF_NO_PARTITION=1
  int ipw_override = 0;        // GRIDPF_IPW=1|2|4 (developer override of the instances-per-wavefront heuristic)
  int wpi_override = 0;        // GRIDPF_WPI=1|2 (developer override of the wavefronts-per-instance heuristic); 1 = deterministic
  int wpi_env = 0;
  bool no_yreg = false;        // GRIDPF_YREG=0: never keep the Ybus blocks in registers
  int stage_max = 2;           // GRIDPF_STAGE=0|1|2: highest static-table staging tier the planner may pick (developer / tests)
  int stage_force = -1;        // GRIDPF_FORCE_STAGE=0|1|2: take that tier whenever it fits the LDS, whatever it costs in residency (experiments)
  int dcf_env = -1;            // GRIDPF_DCF=0|1 (-1: not set)
  int cap_lanes = 0;           // lane buffers are padded to a multiple of 4 lanes (instance groups of a wavefront)
  std::vector<int> lane_mb;    // max live busbars in one substation, per lane
  int init_mb = 1;
  bool plan_valid = false;      // cached launch plan of the whole batch (invalidated by every topology mutation)
  LaunchPlan plan_cached{}, plan_b_cached{};   // plan_b: the split lanes of a mixed batch (sparse_nb == 0: none)
  // profiling
  bool profiling = false;      // per-launch event pairs (gpf_set_profiling(h, 2))
  bool window = false;         // one event pair around a window of launches (gpf_set_profiling(h, 1))
  hipEvent_t win_a = nullptr, win_b = nullptr;
  bool win_marked = false;              // win_b was recorded by gpf_set_profiling(3) behind the last launch of the window
  long long win_launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double acc_ms = 0.0;
  long long acc_launches = 0;

  gpf::Bufs bufs() const {
    gpf::Bufs b{};
    b.inj = inj.p; b.topo = topo.p; b.shunt_bus = shunt_bus.p; b.out = out.p; b.topo_out = topo_out.p;
    b.shunt_bus_out = shunt_bus_out.p; b.line_status = line_status.p; b.status = status.p;
    b.bus_vm = bus_vm.p; b.bus_va = bus_va.p; b.work = work.p; b.work_stride = 0;
    b.chron = chron.p; b.lane_table = lane_table.p; b.lane_offset = lane_offset.p;
    b.lane_scale = has_scale ? lane_scale.p : nullptr;
    b.thermal_limit = thermal_limit.p; b.rho = rho.p; b.overflow_count = overflow_count.p; b.disc_round = disc_round.p;
    b.lane_gen_delta = has_delta ? lane_gen_delta.p : nullptr;
    b.maint = maint.n ? maint.p : nullptr;
    b.cooldown = cooldown.p; b.maint_dur = maint_dur.n ? maint_dur.p : nullptr; b.traj_cool = (traj_cap && traj_cool.n) ? traj_cool.p : nullptr;
    b.topo0 = topo0.p; b.done = done.p; b.episode = episode.p;
    b.traj_rho = traj_cap ? traj_rho.p : nullptr; b.traj_status = traj_cap ? traj_status.p : nullptr; b.traj_cap = traj_cap;
    const bool obs = traj_cap && (traj_what & GPF_TRAJ_OBS);
    b.traj_out = obs ? traj_out.p : nullptr; b.traj_topo = obs ? traj_topo.p : nullptr; b.traj_shb = obs ? traj_shb.p : nullptr;
    b.traj_lstat = obs ? traj_lstat.p : nullptr;
    b.lane_stride = cap_lanes; b.n_real_lanes = n_lanes;
    return b;
  }
};

namespace {

// number of active buses and Newton unknowns of one lane (host mirror of K1's counting)
void count_lane(const gpf_engine* e, const int* topo, const int* shunt_bus, int& nb, int& nj, int& mb) {
  const gpf::GridDev& g = e->g;
  std::vector<unsigned char> act(g.nb_tot, 0), type(g.nb_tot, 0);
  auto mark = [&](int sub, int local) -> int {
    if (local < 1 || local > g.n_busbar) return -1;
    int gb = sub + (local - 1) * g.n_sub;
    act[gb] = 1;
    return gb;
  };
  for (int l = 0; l < g.n_line; ++l) {
    int bo = topo[e->h_line_or_pos[l]], be = topo[e->h_line_ex_pos[l]];
    if (bo >= 1 && be >= 1) { mark(e->h_line_or_sub[l], bo); mark(e->h_line_ex_sub[l], be); }
  }
  for (int i = 0; i < g.n_gen; ++i) {
    int gb = mark(e->h_gen_sub[i], topo[e->h_gen_pos[i]]);
    if (gb >= 0) type[gb] = std::max<unsigned char>(type[gb], e->h_gen_slack[i] ? 2 : 1);
  }
  for (int i = 0; i < g.n_load; ++i) mark(e->h_load_sub[i], topo[e->h_load_pos[i]]);
  for (int i = 0; i < g.n_sto; ++i) mark(e->h_sto_sub[i], topo[e->h_sto_pos[i]]);
  if (shunt_bus)
    for (int i = 0; i < g.n_shunt; ++i) mark(e->h_shunt_sub[i], shunt_bus[i]);
  nb = 0;
  nj = 0;
  for (int b = 0; b < g.nb_tot; ++b) {
    if (!act[b]) continue;
    ++nb;
    if (type[b] == 0) nj += 2;
    else if (type[b] == 1) nj += 1;
  }
  mb = 1;
  for (int s = 0; s < g.n_sub; ++s) {
    int c = 0;
    for (int k = 0; k < g.n_busbar; ++k) c += act[s + k * g.n_sub];
    mb = std::max(mb, c);
  }
}



// flat programs (gridpf_symbolic.hpp: FlatProg) of one graph for the four group widths, in one device buffer; false: upload
// failed.  A graph too large for the 16-bit byte-offset fields gets none (fl[k].n_words == 0: it would not fit the LDS either).
bool upload_flats(const gpf::Symbolic& S, DevArr<int>& buf, gpf::SymDev& D, int lane_opt = 0) {
  D.rslot0 = S.rslot0;
  for (int k = 0; k < 4; ++k) { D.flat[k] = nullptr; D.fl[k] = gpf::FlatDev{}; }
  if (!gpf::flat_fits(S)) return true;
  std::vector<int> all;
  size_t off[4];
  for (int k = 0; k < 4; ++k) {
    // (the lane assignment is a pure function of the graph: engines of the same grid -- every HipBackend copy pool, every test --
    //  share one build per process)
    static std::mutex cache_mu;                                // engines are created from several host threads (one per device)
    static std::unordered_map<std::string, gpf::FlatProg> cache;
    std::string key;
    if (lane_opt > 0) {
      key = std::to_string(S.slot_row.size()) + ":" + std::to_string(S.slot_col.size()) + ":" + std::to_string(S.prog.size()) + ":";
      key.append(reinterpret_cast<const char*>(S.slot_row.data()), S.slot_row.size() * sizeof(int));
      key.append(reinterpret_cast<const char*>(S.slot_col.data()), S.slot_col.size() * sizeof(int));
      key.append(reinterpret_cast<const char*>(S.prog.data()), S.prog.size() * sizeof(int));
      key += "/" + std::to_string(16 << k) + "/" + std::to_string(lane_opt) + "/" + std::to_string(S.gj_lv0) + "/" + std::to_string(S.nslot_lu);
    }
    gpf::FlatProg F;
    bool have = false;
    if (lane_opt > 0) {
      std::lock_guard<std::mutex> lk(cache_mu);
      auto hit = cache.find(key);
      if (hit != cache.end()) { F = hit->second; have = true; }
    }
    if (!have) {
      F = gpf::build_flat(S, 16 << k, lane_opt);               // (the search runs outside the lock: a second thread may repeat it)
      if (lane_opt > 0) {
        std::lock_guard<std::mutex> lk(cache_mu);
        if (cache.size() < 64) cache.emplace(key, F);
      }
    }
    off[k] = all.size();
    all.insert(all.end(), F.words.begin(), F.words.end());
    gpf::FlatDev& f = D.fl[k];
    f.n_fwd = F.n_fwd; f.n_scale = F.n_scale; f.n_scale_rhs = F.n_scale_rhs; f.n_back = F.n_back; f.scale_off = F.scale_off;
    f.back_off = F.back_off; f.rhs_field0 = F.rhs_field0; f.n_words = (int)F.words.size(); f.wave_closed = F.wave_closed ? 1 : 0;
    f.solo_fwd = (int)F.solo_fwd; f.solo_back = (int)F.solo_back;
  }
  if (buf.upload(all.data(), all.size()) != hipSuccess) return false;
  for (int k = 0; k < 4; ++k) D.flat[k] = buf.p + off[k];
  if (g_dry_create) {                           // header-only handle: no device copy; "has flat programs" is what the planner asks (never dereferenced on the host)
    static int dry_words;
    for (int k = 0; k < 4; ++k) D.flat[k] = &dry_words;
  }
  return true;
}

gpf::Symbolic build_symbolic_resident(const gpf::GridDev& g, int n_rows, int n_line, const int* lo, const int* le, bool yb_in_lds);

// Topology class of a lane whose substations are split (gridpf_sparse.hpp: TopoClassDev).  Key = busbar of every line end
// (an open end counts as busbar 1) + which busbars >= 2 carry any element; classes are built on first sight and cached.
// Returns the class id, or -1 (classes disabled / capacity) -> the lane falls back to the NB = n_busbar kernel.
int topo_class_of(gpf_engine* e, const int* topo, const int* shunt_bus) {
  const gpf::GridDev& g = e->g;
  if (e->no_classes || g.n_busbar < 2 || e->classes.size() >= 8192) return -1;
  const int nbb = g.n_busbar;
  std::string key((size_t)2 * g.n_line + (size_t)g.n_sub * nbb, '\0');
  auto bb = [&](int v) { return v >= 2 && v <= nbb ? v : 1; };
  for (int l = 0; l < g.n_line; ++l) {
    key[2 * l] = (char)bb(topo[e->h_line_or_pos[l]]);
    key[2 * l + 1] = (char)bb(topo[e->h_line_ex_pos[l]]);
  }
  char* used = &key[(size_t)2 * g.n_line];                  // [sub][busbar-1]: the busbar carries an element
  auto mark = [&](int sub, int v) { if (v >= 1 && v <= nbb) used[(size_t)sub * nbb + (v - 1)] = 1; };
  for (int l = 0; l < g.n_line; ++l) { mark(e->h_line_or_sub[l], key[2 * l]); mark(e->h_line_ex_sub[l], key[2 * l + 1]); }
  for (int i = 0; i < g.n_gen; ++i) mark(e->h_gen_sub[i], topo[e->h_gen_pos[i]]);
  for (int i = 0; i < g.n_load; ++i) mark(e->h_load_sub[i], topo[e->h_load_pos[i]]);
  for (int i = 0; i < g.n_sto; ++i) mark(e->h_sto_sub[i], topo[e->h_sto_pos[i]]);
  if (shunt_bus) for (int i = 0; i < g.n_shunt; ++i) mark(e->h_shunt_sub[i], shunt_bus[i]);
  std::string elem(used, (size_t)g.n_sub * nbb);                        // busbars that really carry an element
  for (int s_ = 0; s_ < g.n_sub; ++s_) used[(size_t)s_ * nbb] = 1;      // the busbar-1 node always exists
  auto it = e->class_of_key.find(key);
  if (it != e->class_of_key.end()) return it->second;
  // nodes: (sub, busbar 1) -> sub; used busbars >= 2 -> n_sub, n_sub + 1, ...
  std::vector<int> node_of((size_t)g.n_sub * nbb, -1);
  int n_nodes = g.n_sub;
  for (int s_ = 0; s_ < g.n_sub; ++s_) node_of[(size_t)s_ * nbb] = s_;
  for (int s_ = 0; s_ < g.n_sub; ++s_)
    for (int k = 1; k < nbb; ++k) if (used[(size_t)s_ * nbb + k]) node_of[(size_t)s_ * nbb + k] = n_nodes++;
  if (n_nodes > 32767) return -1;
  std::vector<int> lo(g.n_line), le(g.n_line);
  for (int l = 0; l < g.n_line; ++l) {
    lo[l] = node_of[(size_t)e->h_line_or_sub[l] * nbb + (key[2 * l] - 1)];
    le[l] = node_of[(size_t)e->h_line_ex_sub[l] * nbb + (key[2 * l + 1] - 1)];
  }
  gpf::Symbolic S = build_symbolic_resident(g, n_nodes, g.n_line, lo.data(), le.data(), true);
  if (S.nslot > 65535 || !gpf::flat_fits(S)) return -1;
  auto* c = new gpf_engine::TopoClassHost();
  std::vector<int> fi;
  auto puti = [&fi](const int* v, size_t n) { int off = (int)fi.size(); fi.insert(fi.end(), v, v + n); while (fi.size() & 3) fi.push_back(0); return off; };
  const int o_prog = puti(S.prog.data(), S.prog.size());
  std::vector<int> rc(S.nslot_y);
  for (int k = 0; k < S.nslot_y; ++k) rc[k] = S.slot_row[k] | (S.slot_col[k] << 16);
  const int o_rc = puti(rc.data(), rc.size());
  const std::vector<int> upv = gpf::build_upairs(S);
  const int o_up = puti(upv.data(), upv.size());
  const int o_br = puti(S.br_slot.data(), S.br_slot.size());
  const int o_no = puti(node_of.data(), node_of.size());
  if (c->tables.upload(fi.data(), fi.size()) != hipSuccess) { delete c; return -1; }
  c->n_nodes = n_nodes; c->nslot = S.nslot; c->nslot_y = S.nslot_y;
  gpf::SymDev& D = c->dev.sym;
  D = e->sym_dev;                                            // the grid's static blob pointers / offsets
  D.n = S.n; D.nslot = S.nslot; D.nslot_lu = S.nslot_lu; D.nslot_y = S.nslot_y; D.n_levels = S.n_levels; D.back_off = S.back_off; D.back_first = S.back_first;
  D.scale_off = S.scale_off; D.n_scale = S.n_scale; D.n_prog = (int)S.prog.size();
  {   // with every line in service the bus graph of a lane of this class is exactly this graph: connected <=> one component
      // over the nodes that carry an element (an element-only node without a line is an island)
    std::vector<int> comp(n_nodes);
    for (int i = 0; i < n_nodes; ++i) comp[i] = i;
    auto find = [&](int x) { while (comp[x] != x) { comp[x] = comp[comp[x]]; x = comp[x]; } return x; };
    for (int l = 0; l < g.n_line; ++l) comp[find(lo[l])] = find(le[l]);
    int root = -1;
    bool one = true;
    for (int s_ = 0; s_ < g.n_sub && one; ++s_)
      for (int k = 0; k < nbb && one; ++k)
        if (elem[(size_t)s_ * nbb + k]) {
          const int r = find(node_of[(size_t)s_ * nbb + k]);
          if (root < 0) root = r; else one = (r == root);
        }
    D.static_connected = one ? 1 : 0;
  }
  D.prog = c->tables.p + o_prog;
  if (!upload_flats(S, c->flat, D) || !D.fl[3].wave_closed) { c->tables.release(); c->flat.release(); delete c; return -1; }
  c->dev.pair_rc = c->tables.p + o_rc; c->dev.up = c->tables.p + o_up; c->dev.br_slot = c->tables.p + o_br; c->dev.node_of = c->tables.p + o_no;
  D.n_up = (int)upv.size() / 2;
  const int id = (int)e->classes.size();
  e->classes.push_back(c);
  e->class_of_key.emplace(std::move(key), id);
  return id;
}

constexpr size_t LDS_SMALL_LIMIT = 64 * 1024;   // above this Y and J move to an HBM/L2 workspace
constexpr size_t LDS_HARD_LIMIT = 160 * 1024 - 256;   // dynamic LDS budget (a few static bytes: block-wide reductions)

// Symbolic analysis whose Gauss-Jordan tail (Symbolic::gj_lv0) does not cost a resident workgroup per CU.  The 2-wavefront kernels
// of the large grids live on 4 workgroups of ~39 KB per CU: the tail's fill blocks (40 bytes each with the kept DC factors) are
// taken only as far as the count of workgroups that fit stays the same (measured on 118 substations: 52 extra blocks = 3 instead of
// 4 workgroups per CU = -37 %; the allocation is granular and a 40 704-byte workgroup no longer fits four times).  Small grids
// (instance groups / one wavefront, a few hundred bytes of fill) keep the default budget.
size_t gj_workgroups_per_cu(size_t lds_bytes) { return (160 * 1024) / std::max<size_t>((lds_bytes + 512 + 1023) / 1024 * 1024, 1024); }
gpf::Symbolic build_symbolic_resident(const gpf::GridDev& g, int n_rows, int n_line, const int* lo, const int* le, bool yb_in_lds) {
  if (const char* ev = std::getenv("GRIDPF_GJ_BUDGET")) return gpf::build_symbolic(n_rows, n_line, lo, le, 1, std::atoi(ev));   // developer override
  gpf::Symbolic S = gpf::build_symbolic(n_rows, n_line, lo, le);
  if (n_rows < 64) return S;
  auto lds = [&](int nslot) { return gpf::lds_bytes_sparse<1>(g, nslot, yb_in_lds ? S.nslot_y : 0, 0, false, 1, n_rows, true); };
  while (S.nslot > S.nslot_lu && gj_workgroups_per_cu(lds(S.nslot)) < gj_workgroups_per_cu(lds(S.nslot_lu)))
    S = gpf::build_symbolic(n_rows, n_line, lo, le, 1, S.nslot - S.nslot_lu - 1);
  return S;
}

int plan_launch_uncached(gpf_engine* e, int lane0, int n, LaunchPlan& p, LaunchPlan& pb);

// p: the plan of the range (or of its single-busbar lanes when the batch is mixed); pb.sparse_nb != 0: second launch for
// the lanes that have a split substation (both then run from device lane lists)
int plan_launch(gpf_engine* e, int lane0, int n, LaunchPlan& p, LaunchPlan& pb) {
  const bool whole = (lane0 == 0 && n == e->n_lanes);
  if (whole && e->plan_valid) { p = e->plan_cached; pb = e->plan_b_cached; return GPF_OK; }
  int rc = plan_launch_uncached(e, lane0, n, p, pb);
  if (!whole && (p.n_list || pb.n_list)) e->plan_valid = false;     // the device lane lists were overwritten
  if (rc == GPF_OK && whole) { e->plan_cached = p; e->plan_b_cached = pb; e->plan_valid = true; }
  return rc;
}

int plan_launch_uncached(gpf_engine* e, int lane0, int n, LaunchPlan& p, LaunchPlan& pb) {
  pb = LaunchPlan{};
  p = LaunchPlan{};
  p.ipw = 1;
  p.wpi = 1;
  p.minw = 2;
  int mb = 1;
  for (int k = lane0; k < lane0 + n; ++k) mb = std::max(mb, e->lane_mb[k]);
  // kernel S plan of n lanes that all run with NBK busbars per substation block; listed: the lanes come from an index list
  auto plan_sparse = [&](int nbk, int n_l, int lane0_l, bool listed, LaunchPlan& q) -> bool {
    // small grids do not have 64-wide work: several instances share a wavefront (instance groups, gridpf_sparse.hpp).
    // Contiguous sub-ranges must be group-aligned (or end at the padded tail of the lane buffers); lists are padded.
    int ipw = 1;
    if (nbk == 1) {
      ipw = e->ipw_override ? e->ipw_override : (e->g.n_sub <= 8 ? 4 : e->g.n_sub <= 24 ? 2 : 1);
      while (!listed && ipw > 1 && !(lane0_l % ipw == 0 && (n_l % ipw == 0 || lane0_l + n_l == e->n_lanes))) ipw >>= 1;
    }
    auto need = [&](int tier) -> size_t {
      const int wpi_n = (nbk == 1 && ipw == 1) ? (e->wpi_override ? std::min(e->wpi_override, 2) : (e->g.n_sub >= 64 && e->sym_dev.fl[3].wave_closed ? 2 : 1)) : 1;
      // (instance-group kernels stream the flat program from global memory: gridpf_sparse.hpp lu_ac)
      const size_t static_bytes = gpf::stat_bytes(e->sym_dev.so, tier, nbk == 1, ipw > 1 ? 0 : e->sym_dev.fl[gpf::gw_index(64 / ipw * wpi_n)].n_words);
      const bool st = tier > 0;
      const bool dcf = e->dcf != 0;
      return nbk == 1 ? gpf::lds_bytes_sparse<1>(e->g, e->sym.nslot, e->sym.nslot_y, static_bytes, st, ipw, -1, dcf)
           : nbk == 2 ? gpf::lds_bytes_sparse<2>(e->g, e->sym.nslot_lu, e->sym.nslot_y, static_bytes, st, 1, -1, dcf)
                      : gpf::lds_bytes_sparse<3>(e->g, e->sym.nslot_lu, e->sym.nslot_y, static_bytes, st, 1, -1, dcf);
    };
    // stage the static tables + the injection row in LDS only when that does not cost residency: blocks per CU
    // (160 KiB / footprint) must still cover what the launch needs at once, or what the un-staged kernel would get
    const size_t l_gl = need(0);
    const size_t n_blocks = ((size_t)n_l + ipw - 1) / ipw;
    const size_t want = std::min<size_t>(std::min<size_t>((n_blocks + 255) / 256, LDS_HARD_LIMIT / std::max<size_t>(l_gl, 1)), 8);
    int stage = 0;
    const int top_tier = std::min(nbk == 1 ? 2 : 1, e->stage_max);
    for (int tier = top_tier; tier >= 1 && !stage; --tier)
      if (need(tier) <= LDS_HARD_LIMIT && LDS_HARD_LIMIT / need(tier) >= want) stage = tier;
    if (e->stage_force >= 0 && e->stage_force <= top_tier && need(e->stage_force) <= LDS_HARD_LIMIT) stage = e->stage_force;
    if (ipw > 1 && stage != 2) { ipw = 1; stage = 0; for (int tier = top_tier; tier >= 1 && !stage; --tier) if (need(tier) <= LDS_HARD_LIMIT && LDS_HARD_LIMIT / need(tier) >= want) stage = tier; }
    if (e->env_on && ipw == 1) stage = 0;                       // the ENV step kernels exist for tier 0 (and tier 2 with instance groups)
    const size_t l = need(stage);
    if (l > LDS_HARD_LIMIT) return false;
    if (nbk == 1 && !e->sym_dev.flat[0]) return false;           // no flat program: graph beyond the 16-bit slot fields
    q.sparse_nb = nbk;
    q.ipw = ipw;
    q.minw = 2;
    // large grids: phases loop over hundreds of items -> several wavefronts per instance (block-wide barriers)
    // (2 wavefronts per instance need a flat program whose passes keep every destination inside one wavefront -- bitwise
    //  reproducibility; the NB = n_busbar kernels use the level-header program, which is not packed that way: one wavefront
    //  unless GRIDPF_WPI=2 asks for the faster, not bit-reproducible variant)
    q.wpi = (nbk == 1 && ipw == 1) ? (e->wpi_override ? std::min(e->wpi_override, 2) : (e->g.n_sub >= 64 && e->sym_dev.fl[3].wave_closed ? 2 : 1))
          : (nbk == 2 && e->wpi_override == 2) ? 2 : 1;
    if (q.wpi > 1) q.minw = 2;
    q.sparse_stage = stage;
    q.lds = l;
    q.dcf = e->dcf;
    q.yreg = false;
    // Ybus blocks in registers (2 wavefronts per instance, tables in global memory, at most 4 pairs per lane): when the LDS they
    // free holds the factored DC matrix without costing a block per CU, every step of a launch skips the DC assembly + factorisation
    if (nbk == 1 && !listed && ipw == 1 && q.wpi == 2 && stage == 0 && !e->no_yreg && e->dcf_env != 0 && (e->sym.nslot_y - e->g.n_sub) / 2 <= 4 * 128 && e->g.n_sub <= 128) {
      const size_t ly = gpf::lds_bytes_sparse<1>(e->g, e->sym.nslot, 0, 0, false, 1, -1, true);
      if (ly <= LDS_HARD_LIMIT && LDS_HARD_LIMIT / ly >= std::min<size_t>(LDS_HARD_LIMIT / l, want)) { q.yreg = true; q.dcf = 1; q.lds = ly; }
    }
    return true;
  };
  // more than 3 busbars per substation: only through topology classes (the NB = n_busbar block kernels exist for 1..3)
  const bool blocks_ok = e->g.n_busbar <= GPF_MAX_BUSBAR_BLOCKS;
  if (e->g.n_sub * mb <= 32000) {
#ifdef GPF_TIMING
    if (e->work.n < (size_t)e->cap_lanes * gpf::GPF_WORK_ROW) { e->work.release(); HIP_TRY(e->work.alloc((size_t)e->cap_lanes * gpf::GPF_WORK_ROW)); }
#endif
    // lanes with split substations: (1) topology classes -- the single-busbar kernel on the lane's bus-level graph --, else
    // (2) the NB = n_busbar kernel; the lanes without a split keep the plain single-busbar kernel (mixed batch: two launches)
    if (mb > 1 && !e->no_partition) {
      std::vector<int> la, lb, lc;
      bool all_classed = true;
      // ONE topology-class launch serves the whole mixed batch when every lane has a class (the lanes without a split run
      // their own class at the same speed); on small grids the lanes are packed by class into instance groups
      bool everyone = true;
      for (int k = lane0; k < lane0 + n && everyone; ++k) everyone = e->lane_class[k] >= 0;
      const int tc_ipw = e->ipw_override ? e->ipw_override : (e->g.n_sub <= 8 ? 4 : e->g.n_sub <= 24 ? 2 : 1);
      if (everyone) {
        std::vector<int> order(n);
        for (int k = 0; k < n; ++k) order[k] = lane0 + k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return e->lane_class[a] < e->lane_class[b]; });
        for (size_t i = 0; i < order.size();) {
          const int cid = e->lane_class[order[i]];
          size_t j = i;
          while (j < order.size() && e->lane_class[order[j]] == cid) { lb.push_back(order[j]); lc.push_back(cid); ++j; }
          while (lb.size() % tc_ipw) { lb.push_back(e->n_lanes); lc.push_back(cid); }      // ghost lanes complete the group
          i = j;
        }
      } else
      for (int k = lane0; k < lane0 + n; ++k) {
        if (e->lane_mb[k] == 1) la.push_back(k);
        else { lb.push_back(k); lc.push_back(e->lane_class[k]); all_classed &= (e->lane_class[k] >= 0); }
      }
      LaunchPlan qa = p, qb = p;
      bool ok_b;
      if (all_classed) {
        qb.tc = true; qb.sparse_nb = 1; qb.ipw = everyone ? tc_ipw : 1; qb.sparse_stage = 0; qb.minw = 2;
        qb.wpi = qb.ipw > 1 ? 1 : (e->wpi_override ? (e->wpi_override >= 2 ? 2 : 1) : (e->g.n_sub >= 64 ? 2 : 1));
        for (int cid : lc) {
          const auto* c = e->classes[cid];
          qb.tc_rows = std::max(qb.tc_rows, c->n_nodes); qb.tc_nslot = std::max(qb.tc_nslot, c->nslot); qb.tc_nslot_y = std::max(qb.tc_nslot_y, c->nslot_y);
        }
        // DC factors kept across the steps of a launch: alone (every lane has a class: ONE launch, its own parameter block) when
        // that does not cost residency; next to a launch of unsplit lanes the shared parameter block carries that plan's setting
        auto lds_tc = [&](bool dcf_) { return gpf::lds_bytes_sparse<1>(e->g, qb.tc_nslot, qb.tc_nslot_y, 0, false, qb.ipw, qb.tc_rows, dcf_); };
        qb.dcf = everyone ? 0 : e->dcf;
        qb.lds = lds_tc(qb.dcf != 0);
        if (everyone && e->dcf_env != 0) {
          const size_t nblk = (lb.size() + qb.ipw - 1) / qb.ipw, with = lds_tc(true);
          const size_t want_tc = std::min<size_t>(std::min<size_t>((nblk + 255) / 256, LDS_HARD_LIMIT / std::max<size_t>(qb.lds, 1)), 8);
          if (with <= LDS_HARD_LIMIT && LDS_HARD_LIMIT / with >= want_tc) { qb.dcf = 1; qb.lds = with; }
        }
        ok_b = qb.lds <= LDS_HARD_LIMIT;
        if (ok_b && e->d_classes_count != e->classes.size()) {
          std::vector<gpf::TopoClassDev> hc(e->classes.size());
          for (size_t i = 0; i < hc.size(); ++i) hc[i] = e->classes[i]->dev;
          e->d_classes.release();
          HIP_TRY(e->d_classes.upload(hc.data(), hc.size()));
          e->d_classes_count = hc.size();
        }
      } else {
        ok_b = blocks_ok && plan_sparse(e->g.n_busbar, (int)lb.size(), 0, true, qb);
      }
      const bool ok_a = la.empty() || plan_sparse(1, (int)la.size(), 0, true, qa);
      if (ok_a && ok_b) {
        qa.n_list = (int)la.size(); qb.n_list = (int)lb.size();
        while (la.size() % 4) la.push_back(e->n_lanes);             // ghost lane (pristine state, never read back)
        while (lb.size() % 4) { lb.push_back(e->n_lanes); lc.push_back(lc.empty() ? 0 : lc.back()); }
        auto fit = [&](DevArr<int>& d, size_t need_n) -> hipError_t {
          if (d.n >= need_n) return hipSuccess;
          d.release();
          return d.alloc(std::max<size_t>(need_n, (size_t)e->cap_lanes + 8) * 2);
        };
        HIP_TRY(fit(e->list_a, la.size() + 4)); HIP_TRY(fit(e->list_b, lb.size() + 4)); HIP_TRY(fit(e->list_c, lc.size() + 4));
        if (!la.empty()) HIP_TRY(hipMemcpyAsync(e->list_a.p, la.data(), la.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->list_b.p, lb.data(), lb.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->list_c.p, lc.data(), lc.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));                    // the host vectors go out of scope
        qa.list = e->list_a.p; qb.list = e->list_b.p; qb.cls_list = e->list_c.p;
        if (qa.n_list == 0) { p = qb; }                              // every lane is split: one launch
        else { p = qa; pb = qb; }
        return GPF_OK;
      }
    }
    if ((mb == 1 || blocks_ok) && plan_sparse(mb == 1 ? 1 : e->g.n_busbar, n, lane0, false, p)) return GPF_OK;
    if (mb > 1 && !blocks_ok)
      return fail(GPF_E_CAPACITY, "a lane with a split substation has no topology class (GRIDPF_NO_CLASSES / GRIDPF_NO_PARTITION set, or the class "
                                  "could not be built) and the grid has more than 3 busbars per substation: the block kernels cover 1..3");
  }
  return fail(GPF_E_CAPACITY, "grid too large: per-instance LDS footprint of the block-sparse kernel exceeds 160 KiB");
}

int upload_params_s(gpf_engine* e, const gpf::Bufs& b, const LaunchPlan* tc, int dcf) {
  gpf::DevParamsS hp{};
  hp.g = e->g;
  hp.b = b;
  hp.oo = e->oo;
  hp.sym = e->sym_dev;
  hp.classes = e->d_classes.p;
  hp.tc_rows = tc ? tc->tc_rows : 0; hp.tc_nslot = tc ? tc->tc_nslot : 0; hp.tc_nslot_y = tc ? tc->tc_nslot_y : 0;
  hp.dcf = dcf;
  if (e->env_on) {
    gpf::EnvDyn& E = hp.env;
    E.on = 1; E.hold_storage = e->env_hold ? 1 : 0; E.loss_on = e->env_loss_on; E.coeff = e->env_coeff; E.eps_poly = e->rd_eps; E.tol_poly = e->env_tol;
    E.target = e->env_target.p; E.actual = e->env_actual.p; E.prev_p = e->env_prev.p; E.already = e->env_already.p; E.charge = e->env_charge.p;
    E.amount_prev = e->env_amount_prev.p; E.fresh = e->env_fresh.p;
    E.act_redisp = e->env_act_r ? e->env_act_redisp.p : nullptr; E.act_storage = e->env_act_s ? e->env_act_storage.p : nullptr;
    E.limit = e->env_limit.p; E.curt_prev = e->env_curt_prev.p; E.act_curtail = e->env_act_c ? e->env_act_curtail.p : nullptr;
    E.illegal = e->env_illegal.p;
    E.renewable = e->env_has_ren ? e->env_renewable.p : nullptr;
    E.pmin = e->rd_pmin.p; E.pmax = e->rd_pmax.p; E.ramp_up = e->rd_ru.p; E.ramp_down = e->rd_rd.p; E.redispatchable = e->rd_redisp.p;
    E.Emax = e->sto_emax.p; E.Emin = e->sto_emin.p; E.loss = e->sto_loss.p; E.eff_c = e->sto_effc.p; E.eff_d = e->sto_effd.p; E.charge0 = e->sto_charge0.p;
  }
  if (!e->d_params_s) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_params_s), sizeof(gpf::DevParamsS)));
  if (!e->params_s_valid || std::memcmp(&hp, &e->h_params_s, sizeof(hp)) != 0) {
    e->h_params_s = hp;
    HIP_TRY(hipMemcpyAsync(e->d_params_s, &e->h_params_s, sizeof(hp), hipMemcpyHostToDevice, e->stream));
    e->params_s_valid = true;
  }
  return GPF_OK;
}

// the engine's specialised kernels for the launch that follows upload_params_s (nullptr: ahead-of-time kernels).  The literals of the
// specialised kernels ARE this engine's grid: any change of the grid-level part of the parameter block switches them off.
GpfJit* jit_for_launch(gpf_engine* e) {
  if (!e->jit.on) return nullptr;
  const gpf::DevParamsS& hp = e->h_params_s;
  if (std::memcmp(&hp.g, &e->jit_g, sizeof(hp.g)) || std::memcmp(&hp.oo, &e->jit_oo, sizeof(hp.oo)) || std::memcmp(&hp.sym, &e->jit_sym, sizeof(hp.sym))) {
    e->jit.on = false;
    e->jit.message = "the grid-level parameter block changed after gpf_jit_enable: specialised kernels switched off";
    fprintf(stderr, "[gridpf] jit: %s\n", e->jit.message.c_str());
    return nullptr;
  }
  return &e->jit;
}

int prof_begin(gpf_engine* e, hipEvent_t& a, hipEvent_t& b) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t x, y;
    HIP_TRY(hipEventCreate(&x));
    HIP_TRY(hipEventCreate(&y));
    e->ev_pool.emplace_back(x, y);
  }
  a = e->ev_pool[e->ev_used].first;
  b = e->ev_pool[e->ev_used].second;
  ++e->ev_used;
  HIP_TRY(hipEventRecord(a, e->stream));
  return GPF_OK;
}

int drain_events(gpf_engine* e) {
  for (size_t i = 0; i < e->ev_used; ++i) {
    HIP_TRY(hipEventSynchronize(e->ev_pool[i].second));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev_pool[i].first, e->ev_pool[i].second));
    e->acc_ms += ms;
    e->acc_launches += 1;
  }
  e->ev_used = 0;
  return GPF_OK;
}

// env dynamics of lanes [lane0, lane0 + n) back to the state after env.reset(): no dispatch, initial state of charge
int reset_env_state(gpf_engine* e, int lane0, int n) {
  const size_t ng = e->g.n_gen, ns = e->g.n_sto;
  HIP_TRY(hipMemsetAsync(e->env_target.p + lane0 * ng, 0, n * ng * sizeof(float), e->stream));
  HIP_TRY(hipMemsetAsync(e->env_actual.p + lane0 * ng, 0, n * ng * sizeof(float), e->stream));
  HIP_TRY(hipMemsetAsync(e->env_prev.p + lane0 * ng, 0, n * ng * sizeof(float), e->stream));
  HIP_TRY(hipMemsetAsync(e->env_already.p + lane0 * ng, 0, n * ng, e->stream));
  HIP_TRY(hipMemsetAsync(e->env_amount_prev.p + lane0, 0, (size_t)n * sizeof(float), e->stream));
  HIP_TRY(hipMemsetAsync(e->env_curt_prev.p + lane0, 0, (size_t)n * sizeof(float), e->stream));
  HIP_TRY(hipMemsetAsync(e->env_illegal.p + lane0, 0, (size_t)n * sizeof(int), e->stream));
  {
    std::vector<float> ones((size_t)n * ng, 1.0f);
    HIP_TRY(hipMemcpyAsync(e->env_limit.p + lane0 * ng, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  HIP_TRY(hipMemsetAsync(e->env_fresh.p + lane0, 1, (size_t)n, e->stream));
  if (ns) {
    std::vector<float> c((size_t)n * ns);
    for (int k = 0; k < n; ++k) std::copy(e->h_charge0.begin(), e->h_charge0.end(), c.begin() + (size_t)k * ns);
    HIP_TRY(hipMemcpyAsync(e->env_charge.p + lane0 * ns, c.data(), c.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return GPF_OK;
}

int reset_lanes_unchecked(gpf_engine* e, int lane0, int n) {
  HIP_TRY(hipSetDevice(e->device));
  const gpf::GridDev& g = e->g;
  // replicate the pristine rows (host staging keeps this a plain strided copy)
  std::vector<double> inj((size_t)n * g.n_inj);
  std::vector<int> topo((size_t)n * g.dim_topo), sb((size_t)n * g.n_shunt);
  for (int k = 0; k < n; ++k) {
    std::copy(e->h_init_inj.begin(), e->h_init_inj.end(), inj.begin() + (size_t)k * g.n_inj);
    std::copy(e->h_init_topo.begin(), e->h_init_topo.end(), topo.begin() + (size_t)k * g.dim_topo);
    if (g.n_shunt) std::copy(e->h_init_shunt_bus.begin(), e->h_init_shunt_bus.end(), sb.begin() + (size_t)k * g.n_shunt);
  }
  HIP_TRY(hipMemcpyAsync(e->inj.p + (size_t)lane0 * g.n_inj, inj.data(), inj.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->topo.p + (size_t)lane0 * g.dim_topo, topo.data(), topo.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->topo0.p + (size_t)lane0 * g.dim_topo, topo.data(), topo.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  if (g.n_shunt)
    HIP_TRY(hipMemcpyAsync(e->shunt_bus.p + (size_t)lane0 * g.n_shunt, sb.data(), sb.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemsetAsync(e->overflow_count.p + (size_t)lane0 * g.n_line, 0, (size_t)n * g.n_line * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->cooldown.p + (size_t)lane0 * g.n_line, 0, (size_t)n * g.n_line * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->done.p + lane0, 0, (size_t)n, e->stream));
  HIP_TRY(hipMemsetAsync(e->episode.p + (size_t)lane0 * 2, 0, (size_t)n * 2 * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->status.p + (size_t)lane0 * 4, 0xFF, (size_t)n * 4 * sizeof(int), e->stream));
  if (e->env_on) { int rc_e = reset_env_state(e, lane0, n); if (rc_e != GPF_OK) return rc_e; }
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int init_class = topo_class_of(e, e->h_init_topo.data(), g.n_shunt ? e->h_init_shunt_bus.data() : nullptr);
  for (int k = lane0; k < lane0 + n; ++k) {
    e->lane_nb[k] = e->init_nb; e->lane_nj[k] = e->init_nj; e->lane_mb[k] = e->init_mb; e->lane_class[k] = init_class;
    std::copy(e->h_init_topo.begin(), e->h_init_topo.end(), e->h_lane_topo.begin() + (size_t)k * g.dim_topo);
    if (g.n_shunt) std::copy(e->h_init_shunt_bus.begin(), e->h_init_shunt_bus.end(), e->h_lane_sb.begin() + (size_t)k * g.n_shunt);
  }
  e->plan_valid = false;
  return GPF_OK;
}

bool check_range(gpf_engine* e, int lane0, int n) { return e && lane0 >= 0 && n >= 0 && lane0 + n <= e->n_lanes; }

}  // namespace

extern "C" {

const char* gpf_last_error(void) { return g_err.c_str(); }
int gpf_version(void) { return GPF_ABI_VERSION; }

// ---- grid-specialised step kernels (gridpf_jit.hip) ----------------------------------------------------------------------------
namespace {
gpf::DevParamsS jit_block(gpf_engine* e) {
  gpf::DevParamsS hp{};
  hp.g = e->g; hp.oo = e->oo; hp.sym = e->sym_dev;
  return hp;
}
}  // namespace

int gpf_jit_enable(gpf_handle e, const char* src_dir, const char* cache_dir) {
  if (!e) return fail(GPF_E_INVALID, "gpf_jit_enable: null handle");
  (void)hipSetDevice(e->device);
  std::string err;
  if (gpf_jit_configure(e->jit, src_dir, cache_dir, err) != 0) { e->jit.on = false; e->jit.message = err; return fail(GPF_E_UNSUPPORTED, err); }
  const gpf::DevParamsS hp = jit_block(e);
  const std::string header = gpf_jit_header(hp);
  if (!e->jit.can_compile && !gpf_jit_has_aot(e->jit, header)) {       // nothing to load and nothing to compile with: say so now, not launch by launch
    e->jit.on = false;
    e->jit.message = "gpf_jit_enable: no ahead-of-time code objects for this grid (grid2op_amd/_aot) and no compiler at run time: " + e->jit.compile_why;
    return fail(GPF_E_UNSUPPORTED, e->jit.message);
  }
  if (header != e->jit.header) { gpf_jit_release(e->jit); e->jit.header = header; }
  e->jit_g = hp.g; e->jit_oo = hp.oo; e->jit_sym = hp.sym;
  e->jit.on = true;
  return GPF_OK;
}

int gpf_jit_disable(gpf_handle e) {
  if (!e) return fail(GPF_E_INVALID, "gpf_jit_disable: null handle");
  e->jit.on = false;
  return GPF_OK;
}

int gpf_jit_info(gpf_handle e, int64_t* counts, double* seconds, char* text, size_t cap) {
  if (!e) return fail(GPF_E_INVALID, "gpf_jit_info: null handle");
  if (counts) { counts[0] = e->jit.on ? 1 : 0; counts[1] = e->jit.n_compiled; counts[2] = e->jit.n_cached; counts[3] = e->jit.n_failed; counts[4] = e->jit.n_launches; counts[5] = e->jit.n_aot; }
  if (seconds) *seconds = e->jit.seconds;
  if (text && cap) {
    const std::string t = e->jit.variants + (e->jit.message.empty() ? "" : " | " + e->jit.message);
    snprintf(text, cap, "%s", t.c_str());
  }
  return GPF_OK;
}

int gpf_jit_source(gpf_handle e, char* text, size_t cap, size_t* need) {
  if (!e) return fail(GPF_E_INVALID, "gpf_jit_source: null handle");
  const std::string header = gpf_jit_header(jit_block(e));
  if (need) *need = header.size() + 1;
  if (text && cap) snprintf(text, cap, "%s", header.c_str());
  return GPF_OK;
}

int gpf_device_count(int32_t* n_devices) {
  if (!n_devices) return fail(GPF_E_INVALID, "gpf_device_count: null");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *n_devices = n;
  return GPF_OK;
}

int gpf_create(const gpf_grid_desc* d, int32_t n_lanes, int32_t device, gpf_handle* out_h) {
  if (!d || !out_h || n_lanes <= 0) return fail(GPF_E_INVALID, "gpf_create: bad arguments");
  if (d->n_sub <= 0 || d->n_busbar <= 0 || d->n_line < 0 || d->n_gen <= 0) return fail(GPF_E_INVALID, "gpf_create: bad sizes");
  if (d->n_line > 256) return fail(GPF_E_CAPACITY, "gpf_create: n_line > 256 not supported by the step kernel");
  if (d->n_busbar > GPF_MAX_BUSBAR)
    return fail(GPF_E_CAPACITY, "gpf_create: more than 64 busbars per substation (GPF_MAX_BUSBAR)");
  const bool dry = device == GPF_DEVICE_NONE;
  struct DryGuard { bool on; explicit DryGuard(bool d_) : on(d_) { if (on) g_dry_create = true; } ~DryGuard() { if (on) g_dry_create = false; } } dry_guard(dry);
  if (!dry) {
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev == 0)
      return fail(GPF_E_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e0));
    if (device < 0 || device >= ndev) return fail(GPF_E_INVALID, "gpf_create: bad device index");
    HIP_TRY(hipSetDevice(device));
  }
  gpf_engine* e = new gpf_engine();
  e->device = device;
  e->dry = dry;
  {
    const char* np_ = std::getenv("GRIDPF_NO_PARTITION");
    e->no_partition = np_ && np_[0] == '1';
    const char* nc_ = std::getenv("GRIDPF_NO_CLASSES");
    e->no_classes = nc_ && nc_[0] == '1';
    const char* iw = std::getenv("GRIDPF_IPW");
    e->ipw_override = (iw && (iw[0] == '1' || iw[0] == '2' || iw[0] == '4')) ? iw[0] - '0' : 0;
    const char* ww = std::getenv("GRIDPF_WPI");
    e->wpi_override = (ww && (ww[0] == '1' || ww[0] == '2')) ? ww[0] - '0' : 0;
    const char* det = std::getenv("GRIDPF_DETERMINISTIC");
    if (det && det[0] == '1') e->wpi_override = 1;
    e->wpi_env = e->wpi_override;
  }
  e->n_lanes = n_lanes;
  e->cap_lanes = (n_lanes + 7) & ~3;          // >= 4 ghost lanes (pristine state): padding of instance groups / lane lists
  gpf::GridDev& g = e->g;
  g.n_sub = d->n_sub; g.n_busbar = d->n_busbar; g.nb_tot = d->n_sub * d->n_busbar;
  g.n_line = d->n_line; g.n_gen = d->n_gen; g.n_load = d->n_load; g.n_sto = d->n_storage; g.n_shunt = d->n_shunt;
  g.dim_topo = d->dim_topo; g.sn_mva = d->sn_mva; g.inv_sn_mva = 1.0 / d->sn_mva;
  const int nl = g.n_line, ng = g.n_gen, nd = g.n_load, ns = g.n_sto, nsh = g.n_shunt;
  e->oo = gpf::make_offsets(nl, ng, nd, ns, nsh);
  g.n_inj = 2 * ng + 2 * nd + 2 * ns + 2 * nsh;
  g.n_out = 10 * nl + 4 * ng + 4 * nd + 4 * ns + 3 * nsh;
  g.n_chron = 2 * nd + 2 * ng;
  gpf_layout& L = e->layout;
  const gpf::OutOff& o = e->oo;
  L.n_inj = g.n_inj; L.inj_gen_p = o.inj_gen_p; L.inj_gen_vm = o.inj_gen_vm; L.inj_load_p = o.inj_load_p;
  L.inj_load_q = o.inj_load_q; L.inj_storage_p = o.inj_sto_p; L.inj_storage_q = o.inj_sto_q; L.inj_shunt_p = o.inj_sh_p;
  L.inj_shunt_q = o.inj_sh_q;
  L.n_out = g.n_out; L.out_p_or = o.p_or; L.out_q_or = o.q_or; L.out_v_or = o.v_or; L.out_a_or = o.a_or; L.out_theta_or = o.th_or;
  L.out_p_ex = o.p_ex; L.out_q_ex = o.q_ex; L.out_v_ex = o.v_ex; L.out_a_ex = o.a_ex; L.out_theta_ex = o.th_ex;
  L.out_gen_p = o.gen_p; L.out_gen_q = o.gen_q; L.out_gen_v = o.gen_v; L.out_gen_theta = o.gen_th;
  L.out_load_p = o.load_p; L.out_load_q = o.load_q; L.out_load_v = o.load_v; L.out_load_theta = o.load_th;
  L.out_storage_p = o.sto_p; L.out_storage_q = o.sto_q; L.out_storage_v = o.sto_v; L.out_storage_theta = o.sto_th;
  L.out_shunt_p = o.sh_p; L.out_shunt_q = o.sh_q; L.out_shunt_v = o.sh_v;
  L.n_chron = g.n_chron; L.chron_load_p = 0; L.chron_load_q = nd; L.chron_prod_p = 2 * nd; L.chron_prod_v = 2 * nd + ng;
  L.nb_total = g.nb_tot;

  auto cp = [](std::vector<int>& dst, const int32_t* src, int n) { dst.assign(src, src + n); };
  cp(e->h_line_or_sub, d->line_or_sub, nl); cp(e->h_line_ex_sub, d->line_ex_sub, nl);
  cp(e->h_line_or_pos, d->line_or_pos_topo_vect, nl); cp(e->h_line_ex_pos, d->line_ex_pos_topo_vect, nl);
  cp(e->h_gen_sub, d->gen_sub, ng); cp(e->h_gen_pos, d->gen_pos_topo_vect, ng);
  cp(e->h_load_sub, d->load_sub, nd); cp(e->h_load_pos, d->load_pos_topo_vect, nd);
  if (ns) { cp(e->h_sto_sub, d->storage_sub, ns); cp(e->h_sto_pos, d->storage_pos_topo_vect, ns); }
  if (nsh) cp(e->h_shunt_sub, d->shunt_sub, nsh);
  e->h_gen_slack.assign(d->gen_slack, d->gen_slack + ng);
  e->h_br_bdc.assign(d->br_bdc, d->br_bdc + nl);
  e->h_shunt_fact.assign(d->shunt_fact, d->shunt_fact + nsh);
  e->h_init_inj.assign(d->init_inj, d->init_inj + g.n_inj);
  e->h_init_topo.assign(d->init_topo, d->init_topo + g.dim_topo);
  if (nsh) e->h_init_shunt_bus.assign(d->init_shunt_bus, d->init_shunt_bus + nsh);
  // basic validation of the index tables (a wrong table would make the kernels read out of bounds)
  auto in_range = [](const std::vector<int>& v, int hi) { for (int x : v) if (x < 0 || x >= hi) return false; return true; };
  if (!in_range(e->h_line_or_sub, g.n_sub) || !in_range(e->h_line_ex_sub, g.n_sub) || !in_range(e->h_gen_sub, g.n_sub) ||
      !in_range(e->h_load_sub, g.n_sub) || !in_range(e->h_sto_sub, g.n_sub) || !in_range(e->h_shunt_sub, g.n_sub) ||
      !in_range(e->h_line_or_pos, g.dim_topo) || !in_range(e->h_line_ex_pos, g.dim_topo) || !in_range(e->h_gen_pos, g.dim_topo) ||
      !in_range(e->h_load_pos, g.dim_topo) || !in_range(e->h_sto_pos, g.dim_topo)) {
    delete e;
    return fail(GPF_E_INVALID, "gpf_create: index table out of range");
  }

#define UP(arr, src, count)                                                  \
  do {                                                                       \
    hipError_t _e = e->arr.upload(src, (size_t)(count));                     \
    if (_e != hipSuccess) { gpf_destroy(e); return fail(GPF_E_DEVICE, std::string("upload " #arr ": ") + hipGetErrorString(_e)); } \
  } while (0)
#define AL(arr, count)                                                       \
  do {                                                                       \
    hipError_t _e = e->arr.alloc((size_t)(count));                           \
    if (_e != hipSuccess) { gpf_destroy(e); return fail(GPF_E_DEVICE, std::string("alloc " #arr ": ") + hipGetErrorString(_e)); } \
  } while (0)
  UP(sub_vn_kv, d->sub_vn_kv, g.n_sub);
  UP(line_or_sub, d->line_or_sub, nl); UP(line_ex_sub, d->line_ex_sub, nl);
  UP(line_or_pos, d->line_or_pos_topo_vect, nl); UP(line_ex_pos, d->line_ex_pos_topo_vect, nl);
  UP(br_y, d->br_y, 8 * (size_t)nl); UP(br_bdc, d->br_bdc, nl);
  UP(gen_sub, d->gen_sub, ng); UP(gen_pos, d->gen_pos_topo_vect, ng);
  UP(gen_min_q, d->gen_min_q, ng); UP(gen_max_q, d->gen_max_q, ng); UP(gen_slack, d->gen_slack, ng);
  UP(load_sub, d->load_sub, nd); UP(load_pos, d->load_pos_topo_vect, nd);
  UP(sto_sub, d->storage_sub, ns); UP(sto_pos, d->storage_pos_topo_vect, ns);
  UP(shunt_sub, d->shunt_sub, nsh); UP(shunt_fact, d->shunt_fact, nsh);
  UP(d_init_inj, d->init_inj, g.n_inj); UP(d_init_topo, d->init_topo, g.dim_topo); UP(d_init_shunt_bus, d->init_shunt_bus, nsh);
  g.sub_vn_kv = e->sub_vn_kv.p; g.line_or_sub = e->line_or_sub.p; g.line_ex_sub = e->line_ex_sub.p;
  g.line_or_pos = e->line_or_pos.p; g.line_ex_pos = e->line_ex_pos.p; g.br_y = e->br_y.p; g.br_bdc = e->br_bdc.p;
  g.gen_sub = e->gen_sub.p; g.gen_pos = e->gen_pos.p; g.gen_min_q = e->gen_min_q.p; g.gen_max_q = e->gen_max_q.p;
  g.gen_slack = e->gen_slack.p; g.load_sub = e->load_sub.p; g.load_pos = e->load_pos.p; g.sto_sub = e->sto_sub.p;
  g.sto_pos = e->sto_pos.p; g.shunt_sub = e->shunt_sub.p; g.shunt_fact = e->shunt_fact.p;

  const size_t B = (size_t)e->cap_lanes;   // padded: ghost lanes hold the pristine state and are never read back
  AL(inj, B * g.n_inj); AL(topo, B * g.dim_topo); AL(shunt_bus, B * nsh);
  AL(out, B * g.n_out); AL(topo_out, B * g.dim_topo); AL(shunt_bus_out, B * nsh); AL(line_status, B * nl);
  AL(status, B * 4); AL(bus_vm, B * g.nb_tot); AL(bus_va, B * g.nb_tot);
  AL(overflow_count, B * nl); AL(disc_round, B * nl); AL(rho, B * nl); AL(cooldown, B * nl);
  AL(topo0, B * g.dim_topo); AL(done, B); AL(episode, B * 2);
  AL(lane_table, B); AL(lane_offset, B); AL(thermal_limit, nl); AL(tmp_lines, std::max<size_t>(B, 1));
#undef UP
#undef AL
  if (!dry) {
    hipError_t es = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (es != hipSuccess) { gpf_destroy(e); return fail(GPF_E_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(es)); }
  }
  count_lane(e, e->h_init_topo.data(), nsh ? e->h_init_shunt_bus.data() : nullptr, e->init_nb, e->init_nj, e->init_mb);
  e->lane_nb.assign(e->cap_lanes, e->init_nb);
  e->lane_nj.assign(e->cap_lanes, e->init_nj);
  e->lane_mb.assign(e->cap_lanes, e->init_mb);
  e->lane_class.assign(e->cap_lanes, -1);
  e->h_lane_table.assign(e->cap_lanes, 0);
  e->h_lane_offset.assign(e->cap_lanes, 0);
  e->h_lane_forecast.assign(e->cap_lanes, 0);
  e->h_lane_topo.assign((size_t)e->cap_lanes * g.dim_topo, INT_MIN);
  e->h_lane_sb.assign((size_t)e->cap_lanes * std::max(g.n_shunt, 1), INT_MIN);
  {
    // symbolic analysis of the substation graph for the block-sparse kernels (once per grid)
    {
      // + slot layout search (gridpf_symbolic.hpp optimize_slot_layout; GRIDPF_SLOT_OPT=<iterations>, 0 = off); the result is a pure
      // function of the substation graph: one search per process and graph
      static std::mutex mu;
      static std::map<std::string, gpf::Symbolic> cache;
      const char* ev = std::getenv("GRIDPF_SLOT_OPT");
      // (default: grids of more than 24 substations -- one instance per wavefront or two wavefronts per instance, LDS-pipe-bound:
      //  N-1 fan-out on 36 substations +3.9 %, 118 substations +1.2 %; the instance-group kernels of the small grids lose 0.7 %)
      const int slot_opt = ev ? std::atoi(ev) : (g.n_sub > 24 ? 3000 : 0);
      std::string key(reinterpret_cast<const char*>(e->h_line_or_sub.data()), (size_t)nl * sizeof(int));
      key.append(reinterpret_cast<const char*>(e->h_line_ex_sub.data()), (size_t)nl * sizeof(int));
      const char* evb = std::getenv("GRIDPF_GJ_BUDGET");
      key += "/" + std::to_string(g.n_sub) + "/" + std::to_string(slot_opt) + "/" + (evb ? evb : "") + "/" + std::to_string(g.n_gen) + "/" +
             std::to_string(g.n_load) + "/" + std::to_string(g.n_sto) + "/" + std::to_string(g.n_shunt);
      std::lock_guard<std::mutex> lk(mu);
      auto it = cache.find(key);
      if (it == cache.end()) {
        gpf::Symbolic S0 = build_symbolic_resident(g, g.n_sub, nl, e->h_line_or_sub.data(), e->h_line_ex_sub.data(), false);
        it = cache.emplace(key, gpf::optimize_slot_layout(S0, slot_opt)).first;
      }
      e->sym = it->second;
    }
    const gpf::Symbolic& S = e->sym;
    if (S.nslot > 65535 || g.n_sub > 32767) { gpf_destroy(e); return fail(GPF_E_CAPACITY, "grid too large for the 16-bit packed symbolic program"); }
    // the static blob of kernel S (layout: gpf::StatOff): doubles, then ints
    gpf::SymDev& D = e->sym_dev;
    gpf::StatOff& so = D.so;
    std::vector<double> fd;
    std::vector<int> fi;
    auto putd = [&fd](const double* v, size_t n) { int off = (int)fd.size(); fd.insert(fd.end(), v, v + n); if (fd.size() & 1) fd.push_back(0.0); return off; };
    auto puti = [&fi](const int* v, size_t n) { int off = (int)fi.size(); fi.insert(fi.end(), v, v + n); while (fi.size() & 3) fi.push_back(0); return off; };
    so.br_y = putd(d->br_y, (size_t)8 * nl); so.br_bdc = putd(d->br_bdc, nl); so.sub_vn_kv = putd(d->sub_vn_kv, g.n_sub);
    so.shunt_fact = putd(d->shunt_fact, nsh); so.gen_min_q = putd(d->gen_min_q, ng); so.gen_max_q = putd(d->gen_max_q, ng);
    {
      auto vn_of = [&](const int* sub, size_t n) { std::vector<double> v(n); for (size_t i = 0; i < n; ++i) v[i] = d->sub_vn_kv[sub[i]]; return v; };
      std::vector<double> lv((size_t)2 * nl);
      for (int l = 0; l < nl; ++l) { lv[2 * (size_t)l] = d->sub_vn_kv[d->line_or_sub[l]]; lv[2 * (size_t)l + 1] = d->sub_vn_kv[d->line_ex_sub[l]]; }
      so.line_vn = putd(lv.data(), lv.size());
      { std::vector<double> ka(lv.size()); for (size_t i = 0; i < lv.size(); ++i) ka[i] = 1000.0 / (1.7320508075688772935 * lv[i]); so.line_ka = putd(ka.data(), ka.size()); }
      { auto v = vn_of(d->load_sub, nd); so.load_vn = putd(v.data(), v.size()); }
      { auto v = vn_of(d->gen_sub, ng); so.gen_vn = putd(v.data(), v.size()); }
      { auto v = vn_of(d->storage_sub, ns); so.sto_vn = putd(v.data(), v.size()); }
      { auto v = vn_of(d->shunt_sub, nsh); so.shunt_vn = putd(v.data(), v.size()); }
    }
    {   // per-generator totals over the generators of its substation (pfsoln's reactive split when every generator is connected)
      std::vector<double> qmn(g.n_sub, 0.0), qmx(g.n_sub, 0.0), a(ng), b(ng);
      std::vector<int> cn(g.n_sub, 0), nsl(g.n_sub, 0), w(ng);
      for (int i = 0; i < ng; ++i) {
        const int sb = d->gen_sub[i];
        cn[sb] += 1; qmn[sb] += d->gen_min_q[i]; qmx[sb] += d->gen_max_q[i];
        if (d->gen_slack[i]) nsl[sb] += 1;
      }
      for (int i = 0; i < ng; ++i) { const int sb = d->gen_sub[i]; a[i] = qmn[sb]; b[i] = qmx[sb]; w[i] = cn[sb] | (nsl[sb] << 16); }
      so.gen_qmin_tot = putd(a.data(), a.size()); so.gen_qmax_tot = putd(b.data(), b.size());
      e->h_gen_cnt = w;
    }
    so.dc_inv = -1;
    D.dc_inv_g = nullptr;
    if (g.n_sub <= 512 && !std::getenv("GRIDPF_NO_DCINV")) {
      // inverse of the DC matrix of the reference topology (every line in service, reference buses = substations of the slack
      // generators), exactly as K3 assembles it: rows / columns of the reference buses are identity.  Column-major.
      const int n = g.n_sub;
      std::vector<char> is_ref(n, 0);
      for (int i = 0; i < ng; ++i) if (d->gen_slack[i]) is_ref[d->gen_sub[i]] = 1;
      std::vector<double> M((size_t)n * 2 * n, 0.0);
      std::vector<char> touched(n, 0);
      for (int l = 0; l < nl; ++l) {
        const int f = d->line_or_sub[l], t = d->line_ex_sub[l];
        touched[f] = touched[t] = 1;
        if (f == t) continue;
        const double bb = d->br_bdc[l];
        if (!is_ref[f]) M[(size_t)f * 2 * n + f] += bb;
        if (!is_ref[t]) M[(size_t)t * 2 * n + t] += bb;
        if (!is_ref[f] && !is_ref[t]) { M[(size_t)f * 2 * n + t] -= bb; M[(size_t)t * 2 * n + f] -= bb; }
      }
      bool ok = true;
      for (int i = 0; i < n; ++i) { if (is_ref[i]) M[(size_t)i * 2 * n + i] = 1.0; if (!touched[i]) ok = false; M[(size_t)i * 2 * n + n + i] = 1.0; }
      for (int k = 0; k < n && ok; ++k) {                          // Gauss-Jordan, partial pivoting
        int pv = k;
        for (int r = k + 1; r < n; ++r) if (std::fabs(M[(size_t)r * 2 * n + k]) > std::fabs(M[(size_t)pv * 2 * n + k])) pv = r;
        if (!(std::fabs(M[(size_t)pv * 2 * n + k]) > 1e-12)) { ok = false; break; }
        if (pv != k) for (int q = 0; q < 2 * n; ++q) std::swap(M[(size_t)k * 2 * n + q], M[(size_t)pv * 2 * n + q]);
        const double piv = M[(size_t)k * 2 * n + k];
        for (int q = 0; q < 2 * n; ++q) M[(size_t)k * 2 * n + q] /= piv;
        for (int r = 0; r < n; ++r) if (r != k) {
          const double m = M[(size_t)r * 2 * n + k];
          if (m != 0.0) for (int q = 0; q < 2 * n; ++q) M[(size_t)r * 2 * n + q] -= m * M[(size_t)k * 2 * n + q];
        }
      }
      if (ok) {
        // (the matrix is symmetric, its computed inverse only up to rounding: the table is made EXACTLY symmetric, so that the kernels may
        //  read it by rows or by columns, whichever their memory prefers, with identical results)
        std::vector<double> cm((size_t)n * n);
        for (int i = 0; i < n; ++i) for (int k = 0; k < n; ++k) cm[(size_t)k * n + i] = 0.5 * (M[(size_t)i * 2 * n + n + k] + M[(size_t)k * 2 * n + n + i]);
        // small grids: inside the static blob (staged in LDS with it); up to 63 substations (the single-wavefront kernels): a table of
        // its own in global memory, read through L2 (36 substations: 10 KB shared by every lane; +5.7 % at 4 096 lanes, +1.8 % on the
        // N-1 fan-out).  The two-wavefront kernels of the 118-substation grids keep their factored DC matrix in LDS instead: 118
        // dependent L2 reads per bus lane measured -6.8 % against 19 light passes over kept factors.  GRIDPF_DCINV_MAX moves the limit.
        const char* mx = std::getenv("GRIDPF_DCINV_MAX");
        if (n <= 24) so.dc_inv = putd(cm.data(), cm.size());
        else if (n <= (mx ? std::atoi(mx) : 63)) {
          if (e->dc_inv_g.upload(cm.data(), cm.size()) != hipSuccess) { gpf_destroy(e); return fail(GPF_E_DEVICE, "upload dc_inv"); }
          so.dc_inv = -2;
          D.dc_inv_g = e->dc_inv_g.p;
        }
      }
    }
    so.prog = puti(S.prog.data(), S.prog.size());                // 16-byte aligned: level headers are read as int4
    { std::vector<int> rc(S.nslot_y); for (int k = 0; k < S.nslot_y; ++k) rc[k] = S.slot_row[k] | (S.slot_col[k] << 16); so.pair_rc = puti(rc.data(), rc.size()); }
    { const std::vector<int> upv = gpf::build_upairs(S); so.up = puti(upv.data(), upv.size()); D.n_up = (int)upv.size() / 2; }
    so.n_int_hot = (int)fi.size();
    so.line_or_pos = puti(d->line_or_pos_topo_vect, nl); so.line_ex_pos = puti(d->line_ex_pos_topo_vect, nl);
    so.line_or_sub = puti(d->line_or_sub, nl); so.line_ex_sub = puti(d->line_ex_sub, nl);
    so.br_slot = puti(S.br_slot.data(), S.br_slot.size());
    so.gen_pos = puti(d->gen_pos_topo_vect, ng); so.gen_sub = puti(d->gen_sub, ng);
    { std::vector<int> sl(ng); for (size_t i = 0; i < ng; ++i) sl[i] = d->gen_slack[i] ? 1 : 0; so.gen_slack = puti(sl.data(), ng); }
    so.load_pos = puti(d->load_pos_topo_vect, nd); so.load_sub = puti(d->load_sub, nd);
    so.sto_pos = puti(d->storage_pos_topo_vect, ns); so.sto_sub = puti(d->storage_sub, ns);
    so.shunt_sub = puti(d->shunt_sub, nsh);
    so.gen_cnt = puti(e->h_gen_cnt.data(), e->h_gen_cnt.size());
    {
      std::vector<int> pl(g.dim_topo, -1);
      for (int l = 0; l < nl; ++l) { pl[d->line_or_pos_topo_vect[l]] = l; pl[d->line_ex_pos_topo_vect[l]] = l; }
      so.pos_line = puti(pl.data(), pl.size());
    }
    so.n_dbl = (int)fd.size(); so.n_int = (int)fi.size();
    hipError_t eu = e->stat_dbl.upload(fd.data(), fd.size());
    if (eu == hipSuccess) eu = e->stat_int.upload(fi.data(), fi.size());
    if (eu != hipSuccess) { gpf_destroy(e); return fail(GPF_E_DEVICE, std::string("upload static tables: ") + hipGetErrorString(eu)); }
    D.n = S.n; D.nslot = S.nslot; D.nslot_lu = S.nslot_lu; D.nslot_y = S.nslot_y; D.n_levels = S.n_levels; D.back_off = S.back_off; D.back_first = S.back_first;
      {   // connectivity of the static substation graph (all lines in service): lets the kernel skip the label propagation
      std::vector<int> comp(g.n_sub);
      for (int i = 0; i < g.n_sub; ++i) comp[i] = i;
      auto find = [&](int x) { while (comp[x] != x) { comp[x] = comp[comp[x]]; x = comp[x]; } return x; };
      for (int l = 0; l < nl; ++l) comp[find(e->h_line_or_sub[l])] = find(e->h_line_ex_sub[l]);
      int roots = 0;
      for (int i = 0; i < g.n_sub; ++i) roots += (find(i) == i);
      D.static_connected = roots == 1 ? 1 : 0;
    } D.scale_off = S.scale_off; D.n_scale = S.n_scale; D.n_prog = (int)S.prog.size();
    D.stat_dbl = e->stat_dbl.p; D.stat_int = e->stat_int.p; D.prog = e->stat_int.p + so.prog;
    // the grid's own programs get the bank-conflict-aware lane assignment (a local search per pass, ~0.1-1 s once per engine;
    // GRIDPF_LANE_OPT=<iterations> overrides, 0 = sequential assignment); topology classes are built inside a step and skip it
    int lane_opt = 3000;
    if (const char* lo_ = std::getenv("GRIDPF_LANE_OPT")) lane_opt = std::max(0, atoi(lo_));
    if (!upload_flats(S, e->flat_prog, D, lane_opt)) { gpf_destroy(e); return fail(GPF_E_DEVICE, "upload flat programs"); }
  }
  {
    // keep the factored DC matrix in LDS across the steps of a launch when that does not cost residency: the blocks per CU
    // the whole batch needs at once must still fit (GRIDPF_DCF=0|1 overrides)
    const int ipw = e->ipw_override ? e->ipw_override : (g.n_sub <= 8 ? 4 : g.n_sub <= 24 ? 2 : 1);
    const int wpi0 = e->wpi_override ? std::min(e->wpi_override, 2) : (g.n_sub >= 64 ? 2 : 1);
    const size_t stat2 = gpf::stat_bytes(e->sym_dev.so, 2, true, ipw > 1 ? 0 : e->sym_dev.fl[gpf::gw_index(64 / ipw * (ipw == 1 ? wpi0 : 1))].n_words);
    const size_t with = gpf::lds_bytes_sparse<1>(g, e->sym.nslot, e->sym.nslot_y, stat2, true, ipw, -1, true);
    const size_t n_blocks = ((size_t)n_lanes + ipw - 1) / ipw;
    const size_t want = std::min<size_t>((n_blocks + 255) / 256, 8);
    e->dcf = (with <= LDS_HARD_LIMIT && LDS_HARD_LIMIT / with >= want) ? 1 : 0;
    const char* dv = std::getenv("GRIDPF_DCF");
    if (dv && (dv[0] == '0' || dv[0] == '1')) { e->dcf = dv[0] - '0'; e->dcf_env = e->dcf; }
    const char* yv = std::getenv("GRIDPF_YREG");
    e->no_yreg = yv && yv[0] == '0';
    const char* sv_ = std::getenv("GRIDPF_STAGE");
    if (sv_ && sv_[0] >= '0' && sv_[0] <= '2') e->stage_max = sv_[0] - '0';
    const char* kp_ = std::getenv("GRIDPF_KEEP");
    e->keep_enabled = !(kp_ && kp_[0] == '0');
    const char* fs_ = std::getenv("GRIDPF_FORCE_STAGE");
    if (fs_ && fs_[0] >= '0' && fs_[0] <= '2') e->stage_force = fs_[0] - '0';
  }
  if (dry) { *out_h = e; return GPF_OK; }      // header-only handle: gpf_jit_source / gpf_get_plan / gpf_get_layout / gpf_destroy
  HIP_TRY(hipMemsetAsync(e->status.p, 0xFF, B * 4 * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->overflow_count.p, 0, B * nl * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->cooldown.p, 0, B * nl * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->done.p, 0, B, e->stream));
  HIP_TRY(hipMemsetAsync(e->episode.p, 0, B * 2 * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->lane_table.p, 0, B * sizeof(int), e->stream));
  HIP_TRY(hipMemsetAsync(e->lane_offset.p, 0, B * sizeof(int), e->stream));
  {
    std::vector<float> lim(nl, 1e30f);
    HIP_TRY(hipMemcpyAsync(e->thermal_limit.p, lim.data(), nl * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (const char* j = getenv("GRIDPF_JIT")) {
    if (atoi(j) > 0 && gpf_jit_enable(e, nullptr, nullptr) != GPF_OK) fprintf(stderr, "[gridpf] GRIDPF_JIT: %s\n", g_err.c_str());
  }
  *out_h = e;
  int rc = reset_lanes_unchecked(e, 0, e->cap_lanes);
  if (rc != GPF_OK) { gpf_destroy(e); *out_h = nullptr; return rc; }
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_destroy(gpf_handle e) {
  if (!e) return GPF_OK;
  if (e->dry) { for (auto* c : e->classes) delete c; delete e; return GPF_OK; }      // (a header-only handle owns no device resource)
  (void)hipSetDevice(e->device);
  if (e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
  if (e->win_a) { (void)hipEventDestroy(e->win_a); (void)hipEventDestroy(e->win_b); }
  for (auto& pr : e->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  e->sub_vn_kv.release(); e->br_y.release(); e->br_bdc.release(); e->gen_min_q.release(); e->gen_max_q.release();
  e->shunt_fact.release(); e->line_or_sub.release(); e->line_ex_sub.release(); e->line_or_pos.release();
  e->line_ex_pos.release(); e->gen_sub.release(); e->gen_pos.release(); e->load_sub.release(); e->load_pos.release();
  e->sto_sub.release(); e->sto_pos.release(); e->shunt_sub.release(); e->gen_slack.release();
  e->inj.release(); e->bus_vm.release(); e->bus_va.release(); e->work.release(); e->topo.release(); e->shunt_bus.release();
  e->topo_out.release(); e->shunt_bus_out.release(); e->status.release(); e->overflow_count.release(); e->disc_round.release(); e->cooldown.release(); e->keep.release(); e->maint_dur.release(); e->traj_cool.release();
  e->lane_table.release(); e->lane_offset.release(); e->tmp_lines.release(); e->out.release(); e->chron.release();
  e->lane_scale.release(); e->thermal_limit.release(); e->rho.release(); e->line_status.release();
  e->d_init_inj.release(); e->d_init_topo.release(); e->d_init_shunt_bus.release();
  if (e->pin) (void)hipHostFree(e->pin);
  if (e->sim_pin) (void)hipHostFree(e->sim_pin);
  if (e->act_pin) (void)hipHostFree(e->act_pin);
  if (e->res_pin) (void)hipHostFree(e->res_pin);
  if (e->act_up) (void)hipEventDestroy(e->act_up);
  if (e->ptdfb_ev_a) { (void)hipEventDestroy(e->ptdfb_ev_a); (void)hipEventDestroy(e->ptdfb_ev_b); }
  if (e->ptdfb_status_pin) (void)hipHostFree(e->ptdfb_status_pin);
  if (e->ptdfg_info_pin) (void)hipHostFree(e->ptdfg_info_pin);
  if (e->ptdfg_back_pin) (void)hipHostFree(e->ptdfg_back_pin);
  e->ptdfg_hash.release(); e->ptdfg_lane_class.release(); e->ptdfg_first.release(); e->ptdfg_c2b.release(); e->ptdfg_info.release();
  e->maint.release(); e->forecast.release(); e->sim_src.release(); e->sim_rows.release();
  e->env_target.release(); e->env_actual.release(); e->env_prev.release(); e->env_charge.release(); e->env_amount_prev.release();
  e->env_act_storage.p = nullptr; e->env_act_storage.n = 0;       // (an alias into env_act_redisp)
  e->env_act_redisp.release(); e->sto_charge0.release(); e->env_already.release(); e->env_fresh.release();
  e->sto_emax.release(); e->sto_emin.release(); e->sto_loss.release(); e->sto_effc.release(); e->sto_effd.release();
  e->env_limit.release(); e->env_curt_prev.release(); e->env_act_curtail.release(); e->env_renewable.release(); e->env_illegal.release();
  e->rd_pmin.release(); e->rd_pmax.release(); e->rd_ru.release(); e->rd_rd.release(); e->rd_in.release(); e->rd_redisp.release();
  e->rd_u8.release(); e->rd_after.release();
  e->topo0.release(); e->done.release(); e->episode.release(); e->lane_gen_delta.release(); e->traj_rho.release(); e->traj_status.release();
  e->traj_out.release(); e->traj_topo.release(); e->traj_shb.release(); e->traj_lstat.release();
  gpf_jit_release(e->jit);
  if (e->d_params_s) (void)hipFree(e->d_params_s);
  e->stat_dbl.release(); e->dc_inv_g.release();
  e->list_a.release(); e->list_b.release(); e->list_c.release(); e->d_classes.release();
  for (auto* c : e->classes) { c->tables.release(); c->flat.release(); delete c; }
  e->classes.clear();
  e->ptdf_inj_bus.release(); e->ptdf_inj_w.release(); e->ptdf_t.release(); e->ptdf_flow.release();
  e->lodf.release(); e->lodf_worst.release(); e->lodf_inv_cap.release(); e->ptdf_flow_rows.release();
  e->ptdfb_desc.release(); e->ptdfb_order.release(); e->ptdfb_blk_class.release(); e->ptdfb_status.release();
  e->ptdfb_work.release(); e->ptdfb_t.release(); e->ptdfb_lodf.release(); e->ptdfb_inj_w.release();
  e->stat_int.release();
  e->flat_prog.release();
  delete e;
  return GPF_OK;
}

int gpf_get_layout(gpf_handle e, gpf_layout* out) {
  if (!e || !out) return fail(GPF_E_INVALID, "gpf_get_layout: null");
  *out = e->layout;
  return GPF_OK;
}

int gpf_set_deterministic(gpf_handle e, int32_t flag) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_deterministic: null");
  e->wpi_override = flag ? 1 : e->wpi_env;
  e->plan_valid = false;
  return GPF_OK;
}

int gpf_n_lanes(gpf_handle e) { return e ? e->n_lanes : GPF_E_INVALID; }
int gpf_lane_capacity(gpf_handle e) { return e ? e->cap_lanes : GPF_E_INVALID; }

int gpf_set_injections(gpf_handle e, int32_t lane0, int32_t n, const double* inj) {
  if (!check_range(e, lane0, n) || !inj) return fail(GPF_E_INVALID, "gpf_set_injections: bad range");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(e->inj.p + (size_t)lane0 * e->g.n_inj, inj, (size_t)n * e->g.n_inj * sizeof(double),
                         hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));   // the host buffer may be reused by the caller right away
  return GPF_OK;
}

}  // extern "C"
namespace {
// gpf_set_topology.  `engine_rows`: the rows were built by the library itself in its own PINNED block (gpf_simulate_batch: validated
// action items applied to rows that came from the device) -- no validation pass, the uploads are true asynchronous DMA and nothing is
// waited for (the block stays untouched until the stream has drained: the next user of it synchronises first).
int set_topology_rows(gpf_engine* e, int32_t lane0, int32_t n, const int32_t* topo, const int32_t* shunt_bus, bool engine_rows) {
  const gpf::GridDev& g = e->g;
  if (!engine_rows) {
    // validate BOTH arrays before anything is queued: a rejected call leaves the device state and the host mirror untouched
    auto bad_bus = [&g](int v) { return v == 0 || v < -1 || v > g.n_busbar; };
    for (size_t i = 0; i < (size_t)n * g.dim_topo; ++i)
      if (bad_bus(topo[i])) return fail(GPF_E_INVALID, "gpf_set_topology: local bus ids must be -1 or 1..n_busbar");
    if (shunt_bus && g.n_shunt)
      for (size_t i = 0; i < (size_t)n * g.n_shunt; ++i)
        if (bad_bus(shunt_bus[i])) return fail(GPF_E_INVALID, "gpf_set_topology: shunt bus ids must be -1 or 1..n_busbar");
  }
  HIP_TRY(hipMemcpyAsync(e->topo.p + (size_t)lane0 * g.dim_topo, topo, (size_t)n * g.dim_topo * sizeof(int),
                         hipMemcpyHostToDevice, e->stream));
  // the rows an auto-reset restores: copied on the device from the rows just uploaded (not a second trip over PCIe)
  HIP_TRY(hipMemcpyAsync(e->topo0.p + (size_t)lane0 * g.dim_topo, e->topo.p + (size_t)lane0 * g.dim_topo, (size_t)n * g.dim_topo * sizeof(int),
                         hipMemcpyDeviceToDevice, e->stream));
  std::vector<int> sb_host;
  if (shunt_bus && g.n_shunt) {
    HIP_TRY(hipMemcpyAsync(e->shunt_bus.p + (size_t)lane0 * g.n_shunt, shunt_bus, (size_t)n * g.n_shunt * sizeof(int),
                           hipMemcpyHostToDevice, e->stream));
  } else if (g.n_shunt) {
    sb_host.resize((size_t)n * g.n_shunt);
    HIP_TRY(hipMemcpyAsync(sb_host.data(), e->shunt_bus.p + (size_t)lane0 * g.n_shunt, sb_host.size() * sizeof(int),
                           hipMemcpyDeviceToHost, e->stream));
  }
  if (!engine_rows || !sb_host.empty()) HIP_TRY(hipStreamSynchronize(e->stream));
  bool changed = false;
  for (int k = 0; k < n; ++k) {
    const int* t = topo + (size_t)k * g.dim_topo;
    const int* sb = nullptr;
    if (g.n_shunt) sb = shunt_bus ? shunt_bus + (size_t)k * g.n_shunt : sb_host.data() + (size_t)k * g.n_shunt;
    int* mt = e->h_lane_topo.data() + (size_t)(lane0 + k) * g.dim_topo;
    int* ms = e->h_lane_sb.data() + (size_t)(lane0 + k) * std::max(g.n_shunt, 1);
    if (std::memcmp(mt, t, (size_t)g.dim_topo * sizeof(int)) == 0 && (!g.n_shunt || std::memcmp(ms, sb, (size_t)g.n_shunt * sizeof(int)) == 0))
      continue;                                              // re-sent unchanged: counts and topology class stay
    std::memcpy(mt, t, (size_t)g.dim_topo * sizeof(int));
    if (g.n_shunt) std::memcpy(ms, sb, (size_t)g.n_shunt * sizeof(int));
    count_lane(e, t, sb, e->lane_nb[lane0 + k], e->lane_nj[lane0 + k], e->lane_mb[lane0 + k]);
    e->lane_class[lane0 + k] = topo_class_of(e, t, sb);
    changed = true;
  }
  if (changed) e->plan_valid = false;
  return GPF_OK;
}
}  // namespace
extern "C" {

int gpf_set_topology(gpf_handle e, int32_t lane0, int32_t n, const int32_t* topo, const int32_t* shunt_bus) {
  if (!check_range(e, lane0, n) || !topo) return fail(GPF_E_INVALID, "gpf_set_topology: bad range");
  HIP_TRY(hipSetDevice(e->device));
  return set_topology_rows(e, lane0, n, topo, shunt_bus, false);
}

int gpf_get_injections(gpf_handle e, int32_t lane0, int32_t n, double* inj) {
  if (!check_range(e, lane0, n) || !inj) return fail(GPF_E_INVALID, "gpf_get_injections: bad range");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(inj, e->inj.p + (size_t)lane0 * e->g.n_inj, (size_t)n * e->g.n_inj * sizeof(double),
                         hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_topology(gpf_handle e, int32_t lane0, int32_t n, int32_t* topo, int32_t* shunt_bus) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_get_topology: bad range");
  HIP_TRY(hipSetDevice(e->device));
  if (topo)
    HIP_TRY(hipMemcpyAsync(topo, e->topo.p + (size_t)lane0 * e->g.dim_topo, (size_t)n * e->g.dim_topo * sizeof(int),
                           hipMemcpyDeviceToHost, e->stream));
  if (shunt_bus && e->g.n_shunt)
    HIP_TRY(hipMemcpyAsync(shunt_bus, e->shunt_bus.p + (size_t)lane0 * e->g.n_shunt, (size_t)n * e->g.n_shunt * sizeof(int),
                           hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_disconnect_line(gpf_handle e, int32_t lane, int32_t line_id) {
  if (!check_range(e, lane, 1) || line_id < 0 || line_id >= e->g.n_line) return fail(GPF_E_INVALID, "gpf_disconnect_line: bad ids");
  HIP_TRY(hipSetDevice(e->device));
  const int m1 = -1;
  int* row = e->topo.p + (size_t)lane * e->g.dim_topo;
  int* row0 = e->topo0.p + (size_t)lane * e->g.dim_topo;
  HIP_TRY(hipMemcpyAsync(row + e->h_line_or_pos[line_id], &m1, sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(row + e->h_line_ex_pos[line_id], &m1, sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(row0 + e->h_line_or_pos[line_id], &m1, sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(row0 + e->h_line_ex_pos[line_id], &m1, sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->h_lane_topo[(size_t)lane * e->g.dim_topo] = INT_MIN;          // host mirror of the sent topology: unknown from now on
  return GPF_OK;   // removing a branch never increases the bus / unknown counts: capacity bookkeeping unchanged
}

int gpf_reset_lanes(gpf_handle e, int32_t lane0, int32_t n) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_reset_lanes: bad range");
  return reset_lanes_unchecked(e, lane0, n);
}

int gpf_copy_lanes(gpf_handle e, int32_t src, int32_t dst, int32_t n) {
  if (!check_range(e, src, n) || !check_range(e, dst, n)) return fail(GPF_E_INVALID, "gpf_copy_lanes: bad range");
  if (src == dst || n == 0) return GPF_OK;
  if (std::abs(src - dst) < n) return fail(GPF_E_INVALID, "gpf_copy_lanes: overlapping ranges");
  HIP_TRY(hipSetDevice(e->device));
  const gpf::GridDev& g = e->g;
#define CP(arr, stride)                                                                                                     \
  if ((stride) > 0)                                                                                                         \
  HIP_TRY(hipMemcpyAsync(e->arr.p + (size_t)dst * (stride), e->arr.p + (size_t)src * (stride),                              \
                         (size_t)n * (stride) * sizeof(*e->arr.p), hipMemcpyDeviceToDevice, e->stream))
  CP(inj, g.n_inj); CP(topo, g.dim_topo); CP(shunt_bus, g.n_shunt); CP(out, g.n_out); CP(topo_out, g.dim_topo);
  CP(shunt_bus_out, g.n_shunt); CP(line_status, g.n_line); CP(status, 4); CP(bus_vm, g.nb_tot); CP(bus_va, g.nb_tot);
  CP(overflow_count, g.n_line); CP(cooldown, g.n_line); CP(disc_round, g.n_line); CP(rho, g.n_line); CP(topo0, g.dim_topo); CP(done, 1); CP(episode, 2);
  if (e->env_on) {          // the environment's injection dynamics are part of the lane's state (Backend.copy / env.copy keep them)
    CP(env_target, g.n_gen); CP(env_actual, g.n_gen); CP(env_prev, g.n_gen); CP(env_already, g.n_gen); CP(env_limit, g.n_gen);
    CP(env_charge, g.n_sto); CP(env_amount_prev, 1); CP(env_curt_prev, 1); CP(env_fresh, 1); CP(env_illegal, 1);
  }
#undef CP
  for (int k = 0; k < n; ++k) { e->lane_nb[dst + k] = e->lane_nb[src + k]; e->lane_nj[dst + k] = e->lane_nj[src + k]; e->lane_mb[dst + k] = e->lane_mb[src + k]; e->lane_class[dst + k] = e->lane_class[src + k];
    std::copy_n(e->h_lane_topo.begin() + (size_t)(src + k) * g.dim_topo, g.dim_topo, e->h_lane_topo.begin() + (size_t)(dst + k) * g.dim_topo);
    if (g.n_shunt) std::copy_n(e->h_lane_sb.begin() + (size_t)(src + k) * g.n_shunt, g.n_shunt, e->h_lane_sb.begin() + (size_t)(dst + k) * g.n_shunt); }
  e->plan_valid = false;
  return GPF_OK;
}

int gpf_fanout_n1(gpf_handle e, int32_t src, int32_t dst0, int32_t n_out, const int32_t* out_lines) {
  if (!check_range(e, src, 1) || !check_range(e, dst0, n_out) || !out_lines) return fail(GPF_E_INVALID, "gpf_fanout_n1: bad range");
  if (src >= dst0 && src < dst0 + n_out) return fail(GPF_E_INVALID, "gpf_fanout_n1: source inside destination range");
  if (n_out == 0) return GPF_OK;
  HIP_TRY(hipSetDevice(e->device));
  if (e->tmp_lines.n < (size_t)n_out) { e->tmp_lines.release(); HIP_TRY(e->tmp_lines.alloc(n_out)); }
  HIP_TRY(hipMemcpyAsync(e->tmp_lines.p, out_lines, (size_t)n_out * sizeof(int), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(gpf::fanout_kernel, dim3(n_out), dim3(64), 0, e->stream, e->g, e->bufs(), src, dst0, n_out, e->tmp_lines.p);
  HIP_TRY(hipGetLastError());
  if (e->env_on) {                          // the contingency lanes start from the source's injection dynamics
    if (e->sim_src.n < 1) { e->sim_src.release(); HIP_TRY(e->sim_src.alloc(1)); }
    HIP_TRY(hipMemcpyAsync(e->sim_src.p, &src, sizeof(int), hipMemcpyHostToDevice, e->stream));
    gpf::EnvDyn E{};
    E.target = e->env_target.p; E.actual = e->env_actual.p; E.prev_p = e->env_prev.p; E.already = e->env_already.p; E.charge = e->env_charge.p;
    E.amount_prev = e->env_amount_prev.p; E.fresh = e->env_fresh.p; E.limit = e->env_limit.p; E.curt_prev = e->env_curt_prev.p; E.illegal = e->env_illegal.p;
    hipLaunchKernelGGL(gpf::simulate_env_copy_kernel, dim3((unsigned)n_out), dim3(64), 0, e->stream, E, e->g.n_gen, e->g.n_sto, e->sim_src.p, n_out, n_out, dst0);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(e->stream));   // out_lines may be reused by the caller
  for (int k = 0; k < n_out; ++k) { e->lane_nb[dst0 + k] = e->lane_nb[src]; e->lane_nj[dst0 + k] = e->lane_nj[src]; e->lane_mb[dst0 + k] = e->lane_mb[src]; e->lane_class[dst0 + k] = e->lane_class[src];
    e->h_lane_topo[(size_t)(dst0 + k) * e->g.dim_topo] = INT_MIN; }     // a line was forced off on the device: mirror unknown
  e->plan_valid = false;
  return GPF_OK;
}

int gpf_runpf(gpf_handle e, int32_t lane0, int32_t n, int32_t is_dc, int32_t max_iter, double tol_mva) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_runpf: bad range");
  if (n == 0) return GPF_OK;
  HIP_TRY(hipSetDevice(e->device));
  LaunchPlan p, pb;
  int rc = plan_launch(e, lane0, n, p, pb);
  if (rc != GPF_OK) return rc;
  gpf::Bufs b = e->bufs();
  const double tol_pu = tol_mva / e->g.sn_mva;
  hipEvent_t ea = nullptr, eb = nullptr;
  rc = upload_params_s(e, b, p.tc ? &p : (pb.tc ? &pb : nullptr), p.dcf);
  if (rc != GPF_OK) return rc;
  if (e->profiling) { rc = prof_begin(e, ea, eb); if (rc != GPF_OK) return rc; }
  // (measured: forking the second launch onto its own stream costs more in cross-stream events than the overlap gains)
  p.jit = pb.jit = jit_for_launch(e);
  HIP_TRY(gpf_launch_runpf_sparse(p, e->device, e->d_params_s, e->stream, lane0, n, is_dc, max_iter, tol_pu));
  if (pb.sparse_nb) HIP_TRY(gpf_launch_runpf_sparse(pb, e->device, e->d_params_s, e->stream, lane0, n, is_dc, max_iter, tol_pu));
  HIP_TRY(hipGetLastError());
  if (e->profiling) HIP_TRY(hipEventRecord(eb, e->stream));
  if (e->window) { ++e->win_launches; e->win_marked = false; }
  return GPF_OK;
}

// A blocking hipStreamSynchronize wakes up 10-15 us after the stream drained (interrupt + scheduler): a consumer of short launches
// (one 20-step launch of 14 substations is 0.4 ms, a one-lane runpf 30 us) pays that on every synchronisation.  Poll the stream for up
// to ~0.5 ms first (GRIDPF_SYNC_SPIN_US, 0: never), then block: long waits do not burn a host core.
static hipError_t sync_stream_spin(hipStream_t st) {
  static const long spin_us = getenv("GRIDPF_SYNC_SPIN_US") ? atol(getenv("GRIDPF_SYNC_SPIN_US")) : 500;
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) return hipSuccess;
      if (q != hipErrorNotReady) break;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
    }
    (void)hipGetLastError();
  }
  return hipStreamSynchronize(st);
}

// The whole of HipBackend.runpf for ONE lane in a single call: push the lane's injections and topology (apply_action's
// scatter, pandaPowerBackend.py:920-975), solve (runpf -> pp.runpp / rundcpp, :1078-1120), read every result buffer back
// (the getters, :1566-1619) -- everything queued on the stream through one pinned staging block, ONE synchronisation.
int gpf_solve_lane(gpf_handle e, int32_t lane, const double* inj, const int32_t* topo, const int32_t* shunt_bus, int32_t is_dc,
                   int32_t max_iter, double tol_mva, float* out, int32_t* topo_vect, int32_t* shunt_bus_out, uint8_t* line_status,
                   int32_t* status, double* bus_vm, double* bus_va) {
  if (!check_range(e, lane, 1) || !inj || !topo) return fail(GPF_E_INVALID, "gpf_solve_lane: bad arguments");
  const gpf::GridDev& g = e->g;
  if (g.n_shunt && !shunt_bus) return fail(GPF_E_INVALID, "gpf_solve_lane: shunt_bus is required on a grid with shunts");
  auto bad_bus = [&g](int v) { return v == 0 || v < -1 || v > g.n_busbar; };
  for (int i = 0; i < g.dim_topo; ++i) if (bad_bus(topo[i])) return fail(GPF_E_INVALID, "gpf_solve_lane: local bus ids must be -1 or 1..n_busbar");
  for (int i = 0; i < g.n_shunt; ++i) if (bad_bus(shunt_bus[i])) return fail(GPF_E_INVALID, "gpf_solve_lane: shunt bus ids must be -1 or 1..n_busbar");
  HIP_TRY(hipSetDevice(e->device));
  // staging layout (16-byte aligned pieces): in: inj | topo | shunt_bus    out: out | topo_vect | shunt_bus_out | line_status | status | bus_vm | bus_va
  auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t o_inj = 0, o_topo = o_inj + al((size_t)g.n_inj * 8), o_sb = o_topo + al((size_t)g.dim_topo * 4);
  const size_t o_out = o_sb + al((size_t)g.n_shunt * 4), o_tv = o_out + al((size_t)g.n_out * 4), o_sbo = o_tv + al((size_t)g.dim_topo * 4);
  const size_t o_ls = o_sbo + al((size_t)g.n_shunt * 4), o_st = o_ls + al((size_t)g.n_line), o_vm = o_st + 16, o_va = o_vm + al((size_t)g.nb_tot * 8);
  const size_t total = o_va + al((size_t)g.nb_tot * 8);
  if (e->pin_bytes < total) {
    if (e->pin) (void)hipHostFree(e->pin);
    e->pin = nullptr; e->pin_bytes = 0;
    // coherent (fine-grained) mapping: the device reads / writes it uncached, what the gather kernel wrote is in host memory when the
    // stream has drained
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->pin), total, hipHostMallocMapped | hipHostMallocCoherent));
    e->pin_bytes = total;
    e->pin_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->pin_dev), e->pin, 0));
  }
  unsigned char* P = e->pin;
  std::memcpy(P + o_inj, inj, (size_t)g.n_inj * 8);
  std::memcpy(P + o_topo, topo, (size_t)g.dim_topo * 4);
  if (g.n_shunt) std::memcpy(P + o_sb, shunt_bus, (size_t)g.n_shunt * 4);
  hipStream_t st = e->stream;
  // Host bookkeeping of the lane's topology (as gpf_set_topology) and launch planning come FIRST: both can fail (capacity), and a
  // rejected call must leave the device rows and the host mirror untouched -- the old bookkeeping is restored on failure.
  LaunchPlan p, pb;
  {
    int* mt = e->h_lane_topo.data() + (size_t)lane * g.dim_topo;
    int* ms = e->h_lane_sb.data() + (size_t)lane * std::max(g.n_shunt, 1);
    const bool same = std::memcmp(mt, topo, (size_t)g.dim_topo * sizeof(int)) == 0 &&
                      (!g.n_shunt || std::memcmp(ms, shunt_bus, (size_t)g.n_shunt * sizeof(int)) == 0);
    const int o_nb = e->lane_nb[lane], o_nj = e->lane_nj[lane], o_mb = e->lane_mb[lane], o_cls = e->lane_class[lane];
    if (!same) {
      count_lane(e, topo, g.n_shunt ? shunt_bus : nullptr, e->lane_nb[lane], e->lane_nj[lane], e->lane_mb[lane]);
      e->lane_class[lane] = topo_class_of(e, topo, g.n_shunt ? shunt_bus : nullptr);
      e->plan_valid = false;
    }
    int rc = plan_launch(e, lane, 1, p, pb);
    if (rc == GPF_OK) rc = upload_params_s(e, e->bufs(), p.tc ? &p : (pb.tc ? &pb : nullptr), p.dcf);
    if (rc != GPF_OK) {
      e->lane_nb[lane] = o_nb; e->lane_nj[lane] = o_nj; e->lane_mb[lane] = o_mb; e->lane_class[lane] = o_cls;
      e->plan_valid = false;
      return rc;
    }
    if (!same) {
      std::memcpy(mt, topo, (size_t)g.dim_topo * sizeof(int));
      if (g.n_shunt) std::memcpy(ms, shunt_bus, (size_t)g.n_shunt * sizeof(int));
    }
  }
  // GRIDPF_LANE_STAGED=1 (developer A/B): the eleven staged copies of rounds 1 - 4 instead of the two zero-copy dispatches
  static const bool staged = [] { const char* v = std::getenv("GRIDPF_LANE_STAGED"); return v && std::atoi(v) > 0; }();
  const gpf::LaneBlob lb{o_inj, o_topo, o_sb, o_out, o_tv, o_sbo, o_ls, o_st, o_vm, o_va};
  const gpf::Bufs bufs = e->bufs();
  if (staged) {
  HIP_TRY(hipMemcpyAsync(e->inj.p + (size_t)lane * g.n_inj, P + o_inj, (size_t)g.n_inj * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(e->topo.p + (size_t)lane * g.dim_topo, P + o_topo, (size_t)g.dim_topo * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(e->topo0.p + (size_t)lane * g.dim_topo, P + o_topo, (size_t)g.dim_topo * 4, hipMemcpyHostToDevice, st));
  if (g.n_shunt) HIP_TRY(hipMemcpyAsync(e->shunt_bus.p + (size_t)lane * g.n_shunt, P + o_sb, (size_t)g.n_shunt * 4, hipMemcpyHostToDevice, st));
  } else {
    hipLaunchKernelGGL(gpf::lane_scatter_kernel, dim3(1), dim3(256), 0, st, g, bufs, (int)lane, (const unsigned char*)e->pin_dev, lb);
    HIP_TRY(hipGetLastError());
  }
  const double tol_pu = tol_mva / g.sn_mva;
  p.jit = pb.jit = jit_for_launch(e);
  HIP_TRY(gpf_launch_runpf_sparse(p, e->device, e->d_params_s, st, lane, 1, is_dc, max_iter, tol_pu));
  if (pb.sparse_nb) HIP_TRY(gpf_launch_runpf_sparse(pb, e->device, e->d_params_s, st, lane, 1, is_dc, max_iter, tol_pu));
  if (e->window) { ++e->win_launches; e->win_marked = false; }
#define DL1(off, arr, stride, bytes_per) \
  if ((stride) > 0) HIP_TRY(hipMemcpyAsync(P + (off), e->arr.p + (size_t)lane * (stride), (size_t)(stride) * (bytes_per), hipMemcpyDeviceToHost, st))
  if (staged) {
  DL1(o_out, out, g.n_out, 4); DL1(o_tv, topo_out, g.dim_topo, 4); DL1(o_sbo, shunt_bus_out, g.n_shunt, 4); DL1(o_ls, line_status, g.n_line, 1);
  DL1(o_st, status, 4, 4); DL1(o_vm, bus_vm, g.nb_tot, 8); DL1(o_va, bus_va, g.nb_tot, 8);
  } else {
    hipLaunchKernelGGL(gpf::lane_gather_kernel, dim3(1), dim3(256), 0, st, g, bufs, (int)lane, e->pin_dev, lb);
    HIP_TRY(hipGetLastError());
  }
#undef DL1
  HIP_TRY(hipStreamSynchronize(st));          // (a wait this short: the runtime's own active wait beats polling hipStreamQuery by 3 - 5 us)
  if (out) std::memcpy(out, P + o_out, (size_t)g.n_out * 4);
  if (topo_vect) std::memcpy(topo_vect, P + o_tv, (size_t)g.dim_topo * 4);
  if (shunt_bus_out && g.n_shunt) std::memcpy(shunt_bus_out, P + o_sbo, (size_t)g.n_shunt * 4);
  if (line_status) std::memcpy(line_status, P + o_ls, (size_t)g.n_line);
  if (status) std::memcpy(status, P + o_st, 16);
  if (bus_vm) std::memcpy(bus_vm, P + o_vm, (size_t)g.nb_tot * 8);
  if (bus_va) std::memcpy(bus_va, P + o_va, (size_t)g.nb_tot * 8);
  return GPF_OK;
}

int gpf_get_results(gpf_handle e, int32_t lane0, int32_t n, float* out, int32_t* topo_vect, int32_t* shunt_bus,
                    uint8_t* line_status, int32_t* status, double* bus_vm, double* bus_va) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_get_results: bad range");
  HIP_TRY(hipSetDevice(e->device));
  const gpf::GridDev& g = e->g;
#define DL(dst, arr, stride)                                                                                   \
  if ((dst) && (stride) > 0)                                                                                   \
  HIP_TRY(hipMemcpyAsync(dst, e->arr.p + (size_t)lane0 * (stride), (size_t)n * (stride) * sizeof(*e->arr.p),   \
                         hipMemcpyDeviceToHost, e->stream))
  DL(out, out, g.n_out); DL(topo_vect, topo_out, g.dim_topo); DL(shunt_bus, shunt_bus_out, g.n_shunt);
  DL(line_status, line_status, g.n_line); DL(status, status, 4); DL(bus_vm, bus_vm, g.nb_tot); DL(bus_va, bus_va, g.nb_tot);
#undef DL
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

// gpf_get_results into the engine's own PINNED block: the copies are true DMA (a pageable destination is staged through the runtime's
// bounce buffers at a fraction of the PCIe rate), one synchronisation, and the caller reads the rows where they landed -- no second
// host copy.  ptrs[k] = address of piece k inside the block (NULL when not asked for); valid until the next call of this function.
int gpf_get_results_pinned(gpf_handle e, int32_t lane0, int32_t n, int32_t what, void** ptrs) {
  if (!check_range(e, lane0, n) || !ptrs) return fail(GPF_E_INVALID, "gpf_get_results_pinned: bad range");
  HIP_TRY(hipSetDevice(e->device));
  const gpf::GridDev& g = e->g;
  const size_t N = (size_t)n;
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t bytes[8] = {N * g.n_out * 4, N * g.dim_topo * 4, N * g.n_shunt * 4, N * g.n_line, N * 16, N * g.nb_tot * 8, N * g.nb_tot * 8, N * g.n_line * 4};
  size_t off[8], total = 0;
  for (int k = 0; k < 8; ++k) { off[k] = total; if ((what >> k) & 1) total += al(bytes[k]); }
  if (e->res_pin_bytes < total) {
    if (e->res_pin) (void)hipHostFree(e->res_pin);
    e->res_pin = nullptr; e->res_pin_bytes = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->res_pin), total, hipHostMallocDefault));
    e->res_pin_bytes = total;
  }
  const void* src[8] = {e->out.p + (size_t)lane0 * g.n_out, e->topo_out.p + (size_t)lane0 * g.dim_topo, e->shunt_bus_out.p + (size_t)lane0 * g.n_shunt,
                        e->line_status.p + (size_t)lane0 * g.n_line, e->status.p + (size_t)lane0 * 4, e->bus_vm.p + (size_t)lane0 * g.nb_tot,
                        e->bus_va.p + (size_t)lane0 * g.nb_tot, e->rho.p + (size_t)lane0 * g.n_line};
  for (int k = 0; k < 8; ++k) {
    ptrs[k] = nullptr;
    if (!((what >> k) & 1) || bytes[k] == 0) continue;
    HIP_TRY(hipMemcpyAsync(e->res_pin + off[k], src[k], bytes[k], hipMemcpyDeviceToHost, e->stream));
    ptrs[k] = e->res_pin + off[k];
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_upload_chronics(gpf_handle e, int32_t n_tables, int32_t T, const float* data) {
  if (!e || n_tables <= 0 || T <= 0 || !data) return fail(GPF_E_INVALID, "gpf_upload_chronics: bad arguments");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->chron.release();
  e->maint.release();                       // belongs to the previous tables
  e->h_maint.clear(); e->h_hazard.clear();
  e->forecast.release(); e->fc_h = 0;
  HIP_TRY(e->chron.upload(data, (size_t)n_tables * T * e->g.n_chron));
  e->chron_T = T;
  if (n_tables < e->chron_tables) {
    // fewer tables than before: lanes that pointed past the new end fall back to table 0 (the kernel indexes the buffer
    // with lane_table unchecked)
    std::vector<int> lt(e->cap_lanes);
    HIP_TRY(hipMemcpy(lt.data(), e->lane_table.p, lt.size() * sizeof(int), hipMemcpyDeviceToHost));
    bool fix = false;
    for (int& v : lt) if (v < 0 || v >= n_tables) { v = 0; fix = true; }
    if (fix) HIP_TRY(hipMemcpy(e->lane_table.p, lt.data(), lt.size() * sizeof(int), hipMemcpyHostToDevice));
    e->h_lane_table = lt;
  }
  e->chron_tables = n_tables;
  return GPF_OK;
}

}  // extern "C"
namespace {
// maintenance.csv and hazards.csv both force a line out of service while flagged (the environment's "maintenance" / "hazards"
// modifications, Environment/baseEnv.py:2516-2563): the device holds the union of the two tables
int upload_outage_tables(gpf_engine* e) {
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->maint.release();
  e->maint_dur.release();
  if (e->h_maint.empty() && e->h_hazard.empty()) return GPF_OK;
  std::vector<unsigned char> u = e->h_maint.empty() ? e->h_hazard : e->h_maint;
  if (!e->h_maint.empty() && !e->h_hazard.empty()) for (size_t i = 0; i < u.size(); ++i) u[i] = (u[i] || e->h_hazard[i]) ? 1 : 0;
  HIP_TRY(e->maint.upload(u.data(), u.size()));
  // remaining duration of the maintenance / hazard under way at every row (GridValue.get_maintenance_duration_1d / get_hazard_duration_1d
  // while the outage lasts: 3, 2, 1 over a 3-step outage): what BaseEnv._update_time_reconnection_hazards_maintenance raises the line
  // cooldown to (baseEnv.py:2590-2597: the maximum of the two)
  const size_t nl = e->g.n_line, T = (size_t)e->chron_T, nt = (size_t)e->chron_tables;
  std::vector<unsigned short> dur(nt * T * nl, 0);
  auto scan = [&](const std::vector<unsigned char>& tab) {
    if (tab.empty()) return;
    for (size_t k = 0; k < nt; ++k)
      for (size_t l = 0; l < nl; ++l) {
        unsigned run = 0;
        for (size_t t = T; t-- > 0;) {
          run = tab[(k * T + t) * nl + l] ? std::min(run + 1u, 65535u) : 0u;
          unsigned short& d = dur[(k * T + t) * nl + l];
          if (run > d) d = (unsigned short)run;
        }
      }
  };
  scan(e->h_maint); scan(e->h_hazard);
  // (durations given by the caller win: tables that are a WINDOW of longer chronics cannot know how long an outage at their end lasts)
  if (e->h_outage_dur.size() == dur.size()) for (size_t i = 0; i < dur.size(); ++i) if (u[i]) dur[i] = e->h_outage_dur[i];
  HIP_TRY(e->maint_dur.upload(dur.data(), dur.size()));
  return GPF_OK;
}
int set_outage_table(gpf_engine* e, std::vector<unsigned char>& dst, int n_tables, int T, const uint8_t* data, const char* who) {
  if (!data) { dst.clear(); return upload_outage_tables(e); }
  if (n_tables != e->chron_tables || T != e->chron_T)
    return fail(GPF_E_INVALID, std::string(who) + ": shape must match the uploaded chronics tables (n_tables, T)");
  dst.assign(data, data + (size_t)n_tables * T * e->g.n_line);
  return upload_outage_tables(e);
}
}  // namespace
extern "C" {

int gpf_upload_maintenance(gpf_handle e, int32_t n_tables, int32_t T, const uint8_t* data) {
  if (!e) return fail(GPF_E_INVALID, "gpf_upload_maintenance: null");
  return set_outage_table(e, e->h_maint, n_tables, T, data, "gpf_upload_maintenance");
}

int gpf_upload_outage_durations(gpf_handle e, int32_t n_tables, int32_t T, const uint16_t* data) {
  if (!e) return fail(GPF_E_INVALID, "gpf_upload_outage_durations: null");
  if (!data) { e->h_outage_dur.clear(); return upload_outage_tables(e); }
  if (n_tables != e->chron_tables || T != e->chron_T)
    return fail(GPF_E_INVALID, "gpf_upload_outage_durations: shape must match the uploaded chronics tables (n_tables, T)");
  e->h_outage_dur.assign(data, data + (size_t)n_tables * T * e->g.n_line);
  return upload_outage_tables(e);
}

int gpf_upload_hazards(gpf_handle e, int32_t n_tables, int32_t T, const uint8_t* data) {
  if (!e) return fail(GPF_E_INVALID, "gpf_upload_hazards: null");
  return set_outage_table(e, e->h_hazard, n_tables, T, data, "gpf_upload_hazards");
}

int gpf_set_lane_chronics(gpf_handle e, const int32_t* lane_table, const int32_t* lane_offset, const float* lane_scale) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_lane_chronics: null");
  HIP_TRY(hipSetDevice(e->device));
  const size_t B = e->n_lanes;
  if (lane_table) {
    for (size_t k = 0; k < B; ++k)
      if (lane_table[k] < 0 || (e->chron_tables && lane_table[k] >= e->chron_tables)) return fail(GPF_E_INVALID, "lane_table out of range");
    HIP_TRY(hipMemcpyAsync(e->lane_table.p, lane_table, B * sizeof(int), hipMemcpyHostToDevice, e->stream));
    std::copy(lane_table, lane_table + B, e->h_lane_table.begin());
  }
  if (lane_offset) {
    HIP_TRY(hipMemcpyAsync(e->lane_offset.p, lane_offset, B * sizeof(int), hipMemcpyHostToDevice, e->stream));
    std::copy(lane_offset, lane_offset + B, e->h_lane_offset.begin());
    std::fill(e->h_lane_forecast.begin(), e->h_lane_forecast.end(), 0);       // every lane is back on the chronics tables
  }
  if (lane_scale) {
    if (!e->lane_scale.p) {
      HIP_TRY(e->lane_scale.alloc((size_t)e->cap_lanes * 2 * e->g.n_load));
      HIP_TRY(hipMemsetAsync(e->lane_scale.p, 0, (size_t)e->cap_lanes * 2 * e->g.n_load * sizeof(float), e->stream));
    }
    HIP_TRY(hipMemcpyAsync(e->lane_scale.p, lane_scale, B * 2 * e->g.n_load * sizeof(float), hipMemcpyHostToDevice, e->stream));
    e->has_scale = true;
  } else {
    e->has_scale = false;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_set_thermal_limits(gpf_handle e, const float* limit_a) {
  if (!e || !limit_a) return fail(GPF_E_INVALID, "gpf_set_thermal_limits: null");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(e->thermal_limit.p, limit_a, e->g.n_line * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

}  // extern "C"
namespace {
// n_steps env steps of lanes [lane0, lane0 + n) from time index t0 (T rows per chronics table in `b.chron`)
int step_range(gpf_engine* e, const gpf::Bufs& b_in, int lane0, int n, int t0, int T, int n_steps, const gpf_step_opts* o, const char* who,
               bool keep_state = false) {
  LaunchPlan p, pb;
  int rc = plan_launch(e, lane0, n, p, pb);
  if (rc != GPF_OK) return rc;
  if (e->env_on) {
    const int lanes = p.wpi > 1 ? 64 : 64 / std::max(p.ipw, 1);
    if (e->g.n_gen > lanes || e->g.n_sto > lanes || pb.sparse_nb || p.sparse_nb != 1)
      return fail(GPF_E_CAPACITY, std::string(who) + ": the environment dynamics need n_gen and n_storage <= the lanes of an instance (16 / 32 / 64) "
                                  "and a batch that runs as one launch");
  }
  gpf::Bufs b = b_in;
  b.work = e->work.p;                 // (developer timing build: the stamp buffer is allocated by the planner)
  if (n_steps > 1 && pb.sparse_nb)
    return fail(GPF_E_INVALID, std::string(who) + ": multi-step launches need a batch that runs as ONE launch (mixed split / unsplit lanes "
                               "without topology classes run as two): use n_steps = 1");
  if (o->cascade != 0 || b.maint != nullptr) e->dev_topo_dirty = true;     // (lines tripped / taken out by the kernel: see gpf_ptdf_build_batch)
  gpf::StepArgs sa{};
  sa.t = t0; sa.T = T; sa.rebalance_on = o->rebalance > 0.0 ? 1 : 0; sa.rebalance = o->rebalance; sa.cascade = o->cascade;
  sa.is_dc = o->is_dc ? 1 : 0; sa.n_steps = n_steps; sa.auto_reset = o->auto_reset ? 1 : 0; sa.warm_start = o->warm_start ? 1 : 0;
  sa.nb_ts_allowed = o->nb_ts_allowed; sa.max_rounds = o->max_rounds; sa.hard_overflow = o->hard_overflow; sa.soft_overflow = o->soft_overflow;
  sa.nb_ts_reco = o->track_cooldown ? std::max(o->nb_ts_reco, 0) : -1;      // (kernel side: < 0 = the counters are not maintained)
  sa.lane0 = lane0;
  // one-step launches of the engine's own lanes share the topology-derived state of the reference topology (gpf::KeepArgs): single-busbar
  // kernels only (the split lanes of a mixed batch -- topology classes, NB = 2 -- rebuild theirs).  The key is the pristine (ghost) lane's rows.
  if (keep_state && e->keep_enabled && n_steps == 1 && p.sparse_nb == 1 && !p.tc && !o->is_dc) {        // (a DC step builds no Ybus blocks)
    if (!e->keep.p) {
      gpf::KeepArgs k{};
      gpf::keep_layout(k, e->g, e->sym.nslot, e->sym.nslot_y, e->sym_dev.n_up);
      const size_t bytes = 2 * (size_t)k.stride;
      const gpf::OutOff& oo = e->oo;
      if (e->cap_lanes > e->n_lanes && e->keep.alloc(bytes) == hipSuccess) {
        HIP_TRY(hipMemsetAsync(e->keep.p, 0, bytes, e->stream));
        const size_t gl = (size_t)e->n_lanes;                   // first ghost lane: the state every lane is reset to, never mutated
        const int keyed = gpf::KEEP_KEYED;
        for (int v = 0; v < 2; ++v) {
          unsigned char* kb = e->keep.p + (size_t)v * (size_t)k.stride;
          int* kt = reinterpret_cast<int*>(kb) + gpf::KEEP_HDR_INTS;
          HIP_TRY(hipMemcpyAsync(kt, e->topo.p + gl * e->g.dim_topo, (size_t)e->g.dim_topo * sizeof(int), hipMemcpyDeviceToDevice, e->stream));
          if (e->g.n_shunt) {
            HIP_TRY(hipMemcpyAsync(kt + e->g.dim_topo, e->shunt_bus.p + gl * e->g.n_shunt, (size_t)e->g.n_shunt * sizeof(int), hipMemcpyDeviceToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(kb + k.off_kd, e->inj.p + gl * e->g.n_inj + oo.inj_sh_p, 2 * (size_t)e->g.n_shunt * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
          }
          HIP_TRY(hipMemcpyAsync(kb, &keyed, sizeof(int), hipMemcpyHostToDevice, e->stream));
        }
        HIP_TRY(hipStreamSynchronize(e->stream));               // (`keyed` is on this frame)
        k.p = e->keep.p;
        e->keep_args = k;
      } else { (void)hipGetLastError(); e->keep.release(); e->keep_enabled = false; }
    }
    if (e->keep.p) { sa.keep = e->keep_args; sa.keep.launch = ++e->keep_launch; }
  }
  const double tol_pu = o->tol_mva / e->g.sn_mva;
  hipEvent_t ea = nullptr, eb = nullptr;
  rc = upload_params_s(e, b, p.tc ? &p : (pb.tc ? &pb : nullptr), p.dcf);
  if (rc != GPF_OK) return rc;
  if (e->profiling) { rc = prof_begin(e, ea, eb); if (rc != GPF_OK) return rc; }
  p.env = pb.env = e->env_on;
  p.jit = pb.jit = jit_for_launch(e);
  int n_disp = 0;
  HIP_TRY(gpf_launch_step_sparse(p, e->device, e->d_params_s, e->stream, n, o->max_iter, tol_pu, sa, &n_disp));
  if (pb.sparse_nb) HIP_TRY(gpf_launch_step_sparse(pb, e->device, e->d_params_s, e->stream, n, o->max_iter, tol_pu, sa, &n_disp));
  e->n_step_calls += 1; e->n_step_dispatches += n_disp;
  HIP_TRY(hipGetLastError());
  if (e->profiling) HIP_TRY(hipEventRecord(eb, e->stream));
  if (e->window) { ++e->win_launches; e->win_marked = false; }
  return GPF_OK;
}

// _BackendAction.__iadd__ restricted to topology (Action/_backendAction.py:836-919; ValueStore.set_status / change_status / set_val /
// change_val :140-234, _aux_iadd_reconcile_disco_reco :738-765) on one topology row; items = {kind, id, value} triples
void apply_topo_action(const gpf_engine* e, int* row, int* sb, const int* last, const int32_t* items, int n_items) {
  const gpf::GridDev& g = e->g;
  auto old = [&](int pos) { return (last && last[pos] >= 1) ? last[pos] : 1; };
  auto reco = [&](int l) { const int po = e->h_line_or_pos[l], pe = e->h_line_ex_pos[l]; if (row[po] < 0) row[po] = old(po); if (row[pe] < 0) row[pe] = old(pe); };
  auto disco = [&](int l) { row[e->h_line_or_pos[l]] = -1; row[e->h_line_ex_pos[l]] = -1; };
  // III line status: change_status, then set_status (a reconnected end goes back to its last known busbar)
  for (int k = 0; k < n_items; ++k) if (items[3 * k] == GPF_ACT_CHANGE_LINE_STATUS) {
    const int l = items[3 * k + 1];
    if (row[e->h_line_or_pos[l]] > 0 || row[e->h_line_ex_pos[l]] > 0) disco(l); else reco(l);
  }
  for (int k = 0; k < n_items; ++k) if (items[3 * k] == GPF_ACT_SET_LINE_STATUS) {
    const int l = items[3 * k + 1], v = items[3 * k + 2];
    if (v < 0) disco(l); else if (v > 0) reco(l);
  }
  bool any_bus = false;                                         // (the line ends "before" are only needed by rule V)
  for (int k = 0; k < n_items && !any_bus; ++k) any_bus = items[3 * k] == GPF_ACT_CHANGE_BUS || (items[3 * k] == GPF_ACT_SET_BUS && items[3 * k + 2] != 0);
  static thread_local std::vector<int> or_before, ex_before;
  if (any_bus) {
    or_before.resize(g.n_line); ex_before.resize(g.n_line);
    for (int l = 0; l < g.n_line; ++l) { or_before[l] = row[e->h_line_or_pos[l]]; ex_before[l] = row[e->h_line_ex_pos[l]]; }
  }
  // IV change_bus, then set_bus
  bool bus_modif = false;
  for (int k = 0; k < n_items; ++k) if (items[3 * k] == GPF_ACT_CHANGE_BUS) { int& v = row[items[3 * k + 1]]; if (v > 0) v = (1 - v) + 2; bus_modif = true; }
  for (int k = 0; k < n_items; ++k) if (items[3 * k] == GPF_ACT_SET_BUS && items[3 * k + 2] != 0) { row[items[3 * k + 1]] = items[3 * k + 2]; bus_modif = true; }
  // V a line with an open end is open; a line that was open and got a bus on one end is reconnected (other end: last known busbar)
  if (bus_modif)
    for (int l = 0; l < g.n_line; ++l) {
      const int o_ = row[e->h_line_or_pos[l]], x_ = row[e->h_line_ex_pos[l]];
      const bool d_now = or_before[l] == -1 || o_ == -1 || ex_before[l] == -1 || x_ == -1;
      const bool r_now = or_before[l] == -1 && (o_ >= 1 || x_ >= 1);
      if (r_now) reco(l); else if (d_now) disco(l);
    }
  for (int k = 0; k < n_items; ++k) if (items[3 * k] == GPF_ACT_SET_SHUNT_BUS && sb && items[3 * k + 2] != 0) sb[items[3 * k + 1]] = items[3 * k + 2];
}
}  // namespace
extern "C" {

int gpf_step_n(gpf_handle e, int32_t t0, int32_t n_steps, const gpf_step_opts* o) {
  if (!e || !o) return fail(GPF_E_INVALID, "gpf_step_n: null");
  if (n_steps <= 0) return fail(GPF_E_INVALID, "gpf_step_n: n_steps must be positive");
  if (!e->chron.p || e->chron_T <= 0) return fail(GPF_E_INVALID, "gpf_step_n: no chronics uploaded");
  if (e->traj_cap && n_steps > e->traj_cap)
    return fail(GPF_E_INVALID, "gpf_step_n: n_steps exceeds the trajectory buffer (gpf_set_trajectory sizes it; 0 releases it)");
  HIP_TRY(hipSetDevice(e->device));
  int rc = step_range(e, e->bufs(), 0, e->n_lanes, t0, e->chron_T, n_steps, o, "gpf_step_n", true);
  if (rc != GPF_OK) return rc;
  e->traj_valid = e->traj_cap ? n_steps : 0;
  if (e->env_on) {                          // the actions were consumed by this launch (a held storage action stays)
    // (the kernels only read an action buffer whose flag is set -- EnvDyn::act_* is NULL otherwise --, so "consumed" is the flag: no
    //  clearing dispatch behind every launch of an agent that acts at every step)
    e->env_act_r = false;
    if (!e->env_hold) e->env_act_s = false;
    e->env_act_c = false;                   // (the curtailment limits live on in the lanes' state)
  }
  return GPF_OK;
}

int gpf_set_storage_params(gpf_handle e, const double* emax, const double* emin, const double* loss, const double* eff_charge,
                           const double* eff_discharge, const float* charge0, double delta_time_seconds, int32_t activate_loss) {
  if (!e || delta_time_seconds <= 0.0) return fail(GPF_E_INVALID, "gpf_set_storage_params: bad arguments");
  const size_t ns = e->g.n_sto;
  if (ns && (!emax || !emin || !loss || !eff_charge || !eff_discharge || !charge0)) return fail(GPF_E_INVALID, "gpf_set_storage_params: null");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->sto_emax.release(); e->sto_emin.release(); e->sto_loss.release(); e->sto_effc.release(); e->sto_effd.release(); e->sto_charge0.release();
  HIP_TRY(e->sto_emax.upload(emax, ns)); HIP_TRY(e->sto_emin.upload(emin, ns)); HIP_TRY(e->sto_loss.upload(loss, ns));
  HIP_TRY(e->sto_effc.upload(eff_charge, ns)); HIP_TRY(e->sto_effd.upload(eff_discharge, ns)); HIP_TRY(e->sto_charge0.upload(charge0, ns));
  e->h_charge0.assign(charge0, charge0 + ns);
  e->env_coeff = delta_time_seconds / 3600.0;
  e->env_loss_on = activate_loss ? 1 : 0;
  e->sto_ready = true;
  e->params_s_valid = false;
  return GPF_OK;
}

int gpf_set_env_dynamics(gpf_handle e, int32_t on, double tol_poly) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_env_dynamics: null");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->plan_valid = false;
  if (!on) { e->env_on = false; e->params_s_valid = false; return GPF_OK; }
  if (!e->rd_ready) return fail(GPF_E_INVALID, "gpf_set_env_dynamics: call gpf_set_gen_limits first");
  if (e->g.n_sto && !e->sto_ready) return fail(GPF_E_INVALID, "gpf_set_env_dynamics: call gpf_set_storage_params first (the grid has storage units)");
  if (e->g.n_gen > 64 || e->g.n_sto > 64) return fail(GPF_E_CAPACITY, "gpf_set_env_dynamics: more than 64 generators or storage units");
  const size_t B = e->cap_lanes, ng = e->g.n_gen, ns = std::max(e->g.n_sto, 1);
  if (!e->env_target.p) {
    HIP_TRY(e->env_target.alloc(B * ng)); HIP_TRY(e->env_actual.alloc(B * ng)); HIP_TRY(e->env_prev.alloc(B * ng)); HIP_TRY(e->env_already.alloc(B * ng));
    HIP_TRY(e->env_charge.alloc(B * ns)); HIP_TRY(e->env_amount_prev.alloc(B)); HIP_TRY(e->env_fresh.alloc(B));
    // the redispatch and storage action rows of all lanes are ONE allocation ([B][n_gen] | [B][n_storage]): a host agent's two uploads per step are one DMA
    HIP_TRY(e->env_act_redisp.alloc(B * ng + B * ns)); e->env_act_storage.p = e->env_act_redisp.p + B * ng; e->env_act_storage.n = B * ns;
    HIP_TRY(e->env_limit.alloc(B * ng)); HIP_TRY(e->env_curt_prev.alloc(B)); HIP_TRY(e->env_act_curtail.alloc(B * ng)); HIP_TRY(e->env_illegal.alloc(B));
    HIP_TRY(hipMemset(e->env_act_redisp.p, 0, B * ng * sizeof(float))); HIP_TRY(hipMemset(e->env_act_storage.p, 0, B * ns * sizeof(float)));
    HIP_TRY(hipMemset(e->env_charge.p, 0, B * ns * sizeof(float)));
  }
  e->env_on = true;
  e->env_tol = tol_poly > 0.0 ? tol_poly : 1e-2;
  e->env_act_r = e->env_act_s = e->env_act_c = false; e->env_hold = false;
  e->params_s_valid = false;
  return reset_env_state(e, 0, e->cap_lanes);
}

}  // extern "C"
namespace {
// The agents' per-launch actions go through a pinned block of the engine: the caller's (pageable) arrays are copied into it, the
// uploads are true asynchronous DMA behind whatever the stream is doing, and NOTHING is waited for -- a host agent that acts at every
// step paid a staged pageable copy plus a stream synchronisation per step before.  The block is rewritten by the next call, which
// first waits for the event recorded behind this call's uploads (reached long before: they sit in front of the step launch).
// Layout: redispatch [B][n_gen] | storage [B][n_sto] | curtailment [B][n_gen].
int act_pin_begin(gpf_engine* e) {
  const size_t B = e->cap_lanes, ng = e->g.n_gen, ns = std::max(e->g.n_sto, 1), need = B * (2 * ng + ns);     // (the device layout: padded lane count)
  if (e->act_up) HIP_TRY(hipEventSynchronize(e->act_up));
  if (e->act_pin_n < need) {
    if (e->act_pin) (void)hipHostFree(e->act_pin);
    e->act_pin = nullptr; e->act_pin_n = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->act_pin), need * sizeof(float), hipHostMallocDefault));
    std::memset(e->act_pin, 0, need * sizeof(float));           // (the padding lanes' rows travel with the others)
    e->act_pin_n = need;
  }
  if (!e->act_up) HIP_TRY(hipEventCreateWithFlags(&e->act_up, hipEventDisableTiming));
  return GPF_OK;
}
}  // namespace
extern "C" {

int gpf_set_lane_actions(gpf_handle e, const float* redispatch, const float* storage_power, int32_t hold_storage) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_lane_actions: null");
  if (!e->env_on) return fail(GPF_E_INVALID, "gpf_set_lane_actions: the environment dynamics are off (gpf_set_env_dynamics)");
  HIP_TRY(hipSetDevice(e->device));
  const size_t B = e->n_lanes, ng = e->g.n_gen, ns = e->g.n_sto;
  if (redispatch || (storage_power && ns)) {
    const int rc = act_pin_begin(e);
    if (rc != GPF_OK) return rc;
  }
  const size_t Bc = e->cap_lanes, so = Bc * ng;                  // storage rows: behind the (padded) redispatch rows, on the host block as on the device
  if (redispatch) std::memcpy(e->act_pin, redispatch, B * ng * sizeof(float));
  if (storage_power && ns) std::memcpy(e->act_pin + so, storage_power, B * ns * sizeof(float));
  if (redispatch && storage_power && ns) {
    HIP_TRY(hipMemcpyAsync(e->env_act_redisp.p, e->act_pin, (so + B * ns) * sizeof(float), hipMemcpyHostToDevice, e->stream));     // one DMA for both
  } else if (redispatch) {
    HIP_TRY(hipMemcpyAsync(e->env_act_redisp.p, e->act_pin, B * ng * sizeof(float), hipMemcpyHostToDevice, e->stream));
  } else if (storage_power && ns) {
    HIP_TRY(hipMemcpyAsync(e->env_act_storage.p, e->act_pin + so, B * ns * sizeof(float), hipMemcpyHostToDevice, e->stream));
  }
  e->env_act_r = redispatch != nullptr;
  e->env_act_s = storage_power != nullptr && ns > 0;
  e->env_hold = hold_storage != 0;
  if (redispatch || (storage_power && ns)) HIP_TRY(hipEventRecord(e->act_up, e->stream));
  return GPF_OK;
}

int gpf_lane_actions_on_device(gpf_handle e, int32_t redispatch, int32_t storage_power, int32_t curtailment, int32_t hold_storage) {
  if (!e) return fail(GPF_E_INVALID, "gpf_lane_actions_on_device: null");
  if (!e->env_on) return fail(GPF_E_INVALID, "gpf_lane_actions_on_device: the environment dynamics are off (gpf_set_env_dynamics)");
  if (curtailment && !e->env_has_ren) return fail(GPF_E_INVALID, "gpf_lane_actions_on_device: curtailment needs gpf_set_gen_renewable");
  HIP_TRY(hipSetDevice(e->device));
  e->env_act_r = redispatch != 0;
  e->env_act_s = storage_power != 0 && e->g.n_sto > 0;
  e->env_act_c = curtailment != 0;
  e->env_hold = hold_storage != 0;
  return GPF_OK;
}

int gpf_set_gen_renewable(gpf_handle e, const uint8_t* renewable) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_gen_renewable: null");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->env_renewable.release();
  e->env_has_ren = false;
  if (renewable) { HIP_TRY(e->env_renewable.upload(renewable, (size_t)e->g.n_gen)); e->env_has_ren = true; }
  e->params_s_valid = false;
  return GPF_OK;
}

int gpf_set_lane_curtailment(gpf_handle e, const float* limit) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_lane_curtailment: null");
  if (!e->env_on || !e->env_has_ren) return fail(GPF_E_INVALID, "gpf_set_lane_curtailment: needs gpf_set_env_dynamics and gpf_set_gen_renewable");
  HIP_TRY(hipSetDevice(e->device));
  if (!limit) { e->env_act_c = false; return GPF_OK; }
  const size_t n = (size_t)e->n_lanes * e->g.n_gen;
  for (size_t i = 0; i < n; ++i) if (!(limit[i] == -1.0f || (limit[i] >= 0.0f && limit[i] <= 1.0f))) return fail(GPF_E_INVALID, "gpf_set_lane_curtailment: limits are ratios in [0, 1], -1 = no change");
  {
    const int rc = act_pin_begin(e);
    if (rc != GPF_OK) return rc;
  }
  float* stage = e->act_pin + (size_t)e->cap_lanes * (e->g.n_gen + std::max(e->g.n_sto, 1));       // (behind the redispatch | storage rows: act_pin_begin)
  std::memcpy(stage, limit, n * sizeof(float));
  HIP_TRY(hipMemcpyAsync(e->env_act_curtail.p, stage, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipEventRecord(e->act_up, e->stream));
  e->env_act_c = true;
  return GPF_OK;
}

int gpf_get_env_state(gpf_handle e, int32_t lane0, int32_t n, float* target, float* actual, float* prev_p, uint8_t* already_modified,
                      float* charge, float* amount_prev, float* curtail_limit, float* curtail_prev) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_get_env_state: bad range");
  if (!e->env_on) return fail(GPF_E_INVALID, "gpf_get_env_state: the environment dynamics are off");
  HIP_TRY(hipSetDevice(e->device));
  const size_t ng = e->g.n_gen, ns = e->g.n_sto;
#define DLE(dst, arr, stride) if ((dst) && (stride) > 0) HIP_TRY(hipMemcpyAsync(dst, e->arr.p + (size_t)lane0 * (stride), (size_t)n * (stride) * sizeof(*e->arr.p), hipMemcpyDeviceToHost, e->stream))
  DLE(target, env_target, ng); DLE(actual, env_actual, ng); DLE(prev_p, env_prev, ng); DLE(already_modified, env_already, ng);
  DLE(charge, env_charge, ns); DLE(amount_prev, env_amount_prev, 1); DLE(curtail_limit, env_limit, ng); DLE(curtail_prev, env_curt_prev, 1);
#undef DLE
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_env_illegal(gpf_handle e, int32_t lane0, int32_t n, int32_t* count) {
  if (!check_range(e, lane0, n) || !count) return fail(GPF_E_INVALID, "gpf_get_env_illegal: bad range / null");
  if (!e->env_on) return fail(GPF_E_INVALID, "gpf_get_env_illegal: the environment dynamics are off");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(count, e->env_illegal.p + lane0, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_set_env_illegal(gpf_handle e, int32_t lane0, int32_t n, const int32_t* count) {
  if (!check_range(e, lane0, n) || !count) return fail(GPF_E_INVALID, "gpf_set_env_illegal: bad range / null");
  if (!e->env_on) return fail(GPF_E_INVALID, "gpf_set_env_illegal: the environment dynamics are off");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(e->env_illegal.p + lane0, count, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_set_env_state(gpf_handle e, int32_t lane0, int32_t n, const float* target, const float* actual, const float* prev_p,
                      const uint8_t* already_modified, const float* charge, const float* amount_prev, const float* curtail_limit,
                      const float* curtail_prev) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_set_env_state: bad range");
  if (!e->env_on) return fail(GPF_E_INVALID, "gpf_set_env_state: the environment dynamics are off");
  HIP_TRY(hipSetDevice(e->device));
  const size_t ng = e->g.n_gen, ns = e->g.n_sto;
#define ULE(src, arr, stride) if ((src) && (stride) > 0) HIP_TRY(hipMemcpyAsync(e->arr.p + (size_t)lane0 * (stride), src, (size_t)n * (stride) * sizeof(*e->arr.p), hipMemcpyHostToDevice, e->stream))
  ULE(target, env_target, ng); ULE(actual, env_actual, ng); ULE(prev_p, env_prev, ng); ULE(already_modified, env_already, ng);
  ULE(charge, env_charge, ns); ULE(amount_prev, env_amount_prev, 1); ULE(curtail_limit, env_limit, ng); ULE(curtail_prev, env_curt_prev, 1);
#undef ULE
  if (prev_p) HIP_TRY(hipMemsetAsync(e->env_fresh.p + lane0, 0, (size_t)n, e->stream));     // previous set-points given: not a fresh episode
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_upload_forecasts(gpf_handle e, int32_t n_tables, int32_t T, int32_t n_horizons, const float* data) {
  if (!e) return fail(GPF_E_INVALID, "gpf_upload_forecasts: null");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->forecast.release(); e->fc_h = 0;
  if (!data) return GPF_OK;
  if (n_horizons <= 0 || n_tables != e->chron_tables || T != e->chron_T)
    return fail(GPF_E_INVALID, "gpf_upload_forecasts: shape must match the uploaded chronics tables (n_tables, T), n_horizons >= 1");
  HIP_TRY(e->forecast.upload(data, (size_t)n_tables * T * n_horizons * e->g.n_chron));
  e->fc_h = n_horizons;
  return GPF_OK;
}

int gpf_simulate_batch(gpf_handle e, int32_t t_obs, int32_t time_step, int32_t n_src, const int32_t* src_lanes, int32_t n_act,
                       const int32_t* act_off, const int32_t* act_items, const int32_t* last_bus, int32_t dst_lane0,
                       const gpf_step_opts* o) {
  if (!e || !o || !src_lanes || !act_off || n_src <= 0 || n_act <= 0) return fail(GPF_E_INVALID, "gpf_simulate_batch: bad arguments");
  if (!e->chron.p || e->chron_T <= 0) return fail(GPF_E_INVALID, "gpf_simulate_batch: no chronics uploaded");
  if (time_step < 0 || (time_step > 0 && (!e->forecast.p || time_step > e->fc_h)))
    return fail(GPF_E_INVALID, "gpf_simulate_batch: no forecast for that horizon (gpf_upload_forecasts; NoForecastAvailable in the reference)");
  const gpf::GridDev& g = e->g;
  const long long n_dst = (long long)n_src * n_act;
  if (n_dst > e->n_lanes || !check_range(e, dst_lane0, (int)n_dst)) return fail(GPF_E_INVALID, "gpf_simulate_batch: destination range out of bounds");
  for (int b = 0; b < n_src; ++b)
    if (src_lanes[b] < 0 || src_lanes[b] >= e->n_lanes || (src_lanes[b] >= dst_lane0 && src_lanes[b] < dst_lane0 + n_dst))
      return fail(GPF_E_INVALID, "gpf_simulate_batch: source lane out of range or inside the destination range");
  for (int b = 0; b < n_src; ++b)
    if (e->h_lane_forecast[src_lanes[b]])
      return fail(GPF_E_INVALID, "gpf_simulate_batch: a source lane is itself a scratch lane of an earlier gpf_simulate_batch (its chronics cursor is an "
                                 "absolute row, of the forecast tables when it simulated a forecast): chained simulate is not supported -- simulate from "
                                 "the environment's lane, or send gpf_set_lane_chronics again");
  const int n_items_total = act_off[n_act];
  if (act_off[0] != 0 || (n_items_total > 0 && !act_items)) return fail(GPF_E_INVALID, "gpf_simulate_batch: bad action offsets");
  for (int k = 0; k < n_act; ++k) if (act_off[k + 1] < act_off[k]) return fail(GPF_E_INVALID, "gpf_simulate_batch: bad action offsets");
  for (int q = 0; q < n_items_total; ++q) {
    const int kind = act_items[3 * q], id = act_items[3 * q + 1], v = act_items[3 * q + 2];
    const bool pos_kind = kind == GPF_ACT_SET_BUS || kind == GPF_ACT_CHANGE_BUS, line_kind = kind == GPF_ACT_SET_LINE_STATUS || kind == GPF_ACT_CHANGE_LINE_STATUS;
    if (!(pos_kind || line_kind || kind == GPF_ACT_SET_SHUNT_BUS) || id < 0 || (pos_kind && id >= g.dim_topo) || (line_kind && id >= g.n_line) ||
        (kind == GPF_ACT_SET_SHUNT_BUS && id >= g.n_shunt) || ((kind == GPF_ACT_SET_BUS || kind == GPF_ACT_SET_SHUNT_BUS) && (v < -1 || v > g.n_busbar)) ||
        (kind == GPF_ACT_CHANGE_BUS && g.n_busbar != 2))
      return fail(GPF_E_INVALID, "gpf_simulate_batch: bad action item (kind, id or bus; change_bus needs exactly 2 busbars per substation)");
  }
  HIP_TRY(hipSetDevice(e->device));
  static const bool sim_timing = getenv("GRIDPF_SIM_TIMING") != nullptr;        // developer: stage times of the call on stderr
  auto now_us = [] { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3; };
  const double tm0 = sim_timing ? now_us() : 0.0;
  // 1. the source lanes' topology / shunt rows as they are on the device NOW (trips and maintenance included) -> host
  const int w = g.dim_topo + g.n_shunt;
  if (e->sim_src.n < (size_t)n_src) { e->sim_src.release(); HIP_TRY(e->sim_src.alloc((size_t)n_src)); }
  if (e->sim_rows.n < (size_t)n_src * w) { e->sim_rows.release(); HIP_TRY(e->sim_rows.alloc((size_t)n_src * w)); }
  HIP_TRY(hipMemcpyAsync(e->sim_src.p, src_lanes, (size_t)n_src * sizeof(int), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(gpf::gather_topo_kernel, dim3(n_src), dim3(64), 0, e->stream, e->g, e->bufs(), e->sim_src.p, n_src, e->sim_rows.p);
  HIP_TRY(hipGetLastError());
  // one pinned host block for the rows that cross PCIe in this call (true DMA both ways, no zero-filled vectors): the gathered source
  // rows, then the candidate topology / shunt rows.  It is only rewritten after the synchronisation below, i.e. when the uploads of
  // the previous call have long been consumed.
  const size_t n_rows = (size_t)n_src * w, n_topo = (size_t)n_dst * g.dim_topo, n_sb = (size_t)n_dst * std::max(g.n_shunt, 1);
  if (e->sim_pin_n < n_rows + n_topo + n_sb) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->sim_pin) (void)hipHostFree(e->sim_pin);
    e->sim_pin = nullptr; e->sim_pin_n = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->sim_pin), (n_rows + n_topo + n_sb) * sizeof(int), hipHostMallocDefault));
    e->sim_pin_n = n_rows + n_topo + n_sb;
  }
  int* const rows = e->sim_pin;
  HIP_TRY(hipMemcpyAsync(rows, e->sim_rows.p, n_rows * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const double tm1 = sim_timing ? now_us() : 0.0;
  // 2. candidate topologies on the host (the launch planner needs them anyway: busbars per substation, topology classes)
  int* const topo = e->sim_pin + n_rows;
  int* const sb = topo + n_topo;
  // Scheduled maintenance ahead of the observation (_ObsEnv.init, Environment/_obsEnv.py:361-385 with
  // BaseEnv._update_vector_with_timestep, baseEnv.py:4768-4825): a forecast `time_step` >= 1 steps ahead has the lines out whose
  // NEXT maintenance (the one obs.time_next_maintenance / duration_next_maintenance describe: the first flagged row from the
  // observation's row on) covers row idx + time_step -- it begins there (first_ts_maintenance) or is under way
  // (still_in_maintenance).  Hazards are not forecast by the reference.  The line goes out BEFORE the candidate action is applied.
  std::vector<std::vector<int>> maint_out(n_src);
  if (time_step >= 1 && !e->h_maint.empty()) {
    const int T = e->chron_T;
    for (int b = 0; b < n_src; ++b) {
      const int src = src_lanes[b];
      int idx = (t_obs + e->h_lane_offset[src]) % T;
      if (idx < 0) idx += T;
      const unsigned char* M = e->h_maint.data() + (size_t)e->h_lane_table[src] * T * g.n_line;
      for (int l = 0; l < g.n_line; ++l) {
        int s_ = idx;
        while (s_ < T && !M[(size_t)s_ * g.n_line + l]) ++s_;          // start of the next maintenance (idx itself: under way)
        const int tgt = idx + time_step;
        if (s_ >= T || tgt < s_ || tgt >= T) continue;
        bool in = true;
        for (int r = s_; r <= tgt && in; ++r) in = M[(size_t)r * g.n_line + l] != 0;
        if (in) maint_out[b].push_back(l);
      }
    }
  }
  for (int b = 0; b < n_src; ++b)
    for (int k = 0; k < n_act; ++k) {
      int* row = topo + ((size_t)b * n_act + k) * g.dim_topo;
      int* srow = sb + ((size_t)b * n_act + k) * std::max(g.n_shunt, 1);
      std::memcpy(row, rows + (size_t)b * w, (size_t)g.dim_topo * sizeof(int));
      if (g.n_shunt) std::memcpy(srow, rows + (size_t)b * w + g.dim_topo, (size_t)g.n_shunt * sizeof(int));
      for (int l : maint_out[b]) { row[e->h_line_or_pos[l]] = -1; row[e->h_line_ex_pos[l]] = -1; }
      apply_topo_action(e, row, g.n_shunt ? srow : nullptr, last_bus ? last_bus + (size_t)b * g.dim_topo : nullptr,
                        act_items + 3 * (size_t)act_off[k], act_off[k + 1] - act_off[k]);
    }
  const double tm2 = sim_timing ? now_us() : 0.0;
  int rc = set_topology_rows(e, dst_lane0, (int)n_dst, topo, g.n_shunt ? sb : nullptr, true);
  if (rc != GPF_OK) return rc;
  const double tm3 = sim_timing ? now_us() : 0.0;
  // 3. everything else of the source lanes + the chronics / forecast row to step on, on the device
  const bool fc = time_step > 0;
  const int T_eff = fc ? e->chron_T * e->fc_h : e->chron_T;
  if (e->has_scale && !e->lane_scale.p) return fail(GPF_E_INVALID, "gpf_simulate_batch: internal (lane_scale)");
  hipLaunchKernelGGL(gpf::simulate_prepare_kernel, dim3((unsigned)n_dst), dim3(64), 0, e->stream, e->g, e->bufs(), e->sim_src.p, n_act, (int)n_dst,
                     dst_lane0, t_obs, e->chron_T, fc ? e->fc_h : 1, time_step, e->lane_table.p, e->lane_offset.p,
                     e->has_scale ? e->lane_scale.p : nullptr, e->has_delta ? e->lane_gen_delta.p : nullptr);
  HIP_TRY(hipGetLastError());
  for (long long q = 0; q < n_dst; ++q) {                     // host mirror of what the kernel writes
    const int src = src_lanes[q / n_act];
    int idx = (t_obs + e->h_lane_offset[src]) % e->chron_T;
    if (idx < 0) idx += e->chron_T;
    e->h_lane_table[dst_lane0 + q] = e->h_lane_table[src];
    e->h_lane_offset[dst_lane0 + q] = time_step == 0 ? idx : (fc ? e->fc_h : 1) * idx + (time_step - 1);
    e->h_lane_forecast[dst_lane0 + q] = 1;          // its cursor is an ABSOLUTE row (of the forecast tables when time_step > 0), not an offset to t
  }
  // with the injection dynamics on, the scratch lanes start from their source's dispatch / storage / curtailment state and take
  // ONE do-nothing step of the dynamics on the forecast (the candidates are topology actions: no redispatch / storage part)
  struct ActGuard {
    gpf_engine* e; bool r, s, c;
    explicit ActGuard(gpf_engine* e_) : e(e_), r(e_->env_act_r), s(e_->env_act_s), c(e_->env_act_c) { e->env_act_r = e->env_act_s = e->env_act_c = false; }
    ~ActGuard() { e->env_act_r = r; e->env_act_s = s; e->env_act_c = c; }
  } act_guard(e);
  if (e->env_on) {
    gpf::EnvDyn E{};
    E.target = e->env_target.p; E.actual = e->env_actual.p; E.prev_p = e->env_prev.p; E.already = e->env_already.p; E.charge = e->env_charge.p;
    E.amount_prev = e->env_amount_prev.p; E.fresh = e->env_fresh.p; E.limit = e->env_limit.p; E.curt_prev = e->env_curt_prev.p; E.illegal = e->env_illegal.p;
    hipLaunchKernelGGL(gpf::simulate_env_copy_kernel, dim3((unsigned)n_dst), dim3(64), 0, e->stream, E, g.n_gen, g.n_sto, e->sim_src.p, n_act,
                       (int)n_dst, dst_lane0);
    HIP_TRY(hipGetLastError());
  }
  // 4. ONE step of the destination range on the forecast tables (no trajectory rows: these are scratch lanes)
  gpf::Bufs b = e->bufs();
  if (fc) b.chron = e->forecast.p;
  b.maint = nullptr;
  b.maint_dur = nullptr;               // (the cursor of a scratch lane is a row of the FORECAST tables: the outage tables have no such rows)
  b.traj_rho = nullptr; b.traj_status = nullptr; b.traj_out = nullptr; b.traj_topo = nullptr; b.traj_shb = nullptr; b.traj_lstat = nullptr; b.traj_cap = 0;
  gpf_step_opts oo = *o;
  oo.auto_reset = 0; oo.warm_start = 0;
  oo.track_cooldown = 0;               // one look-ahead step: the cooldowns copied from the source lanes stand
  const double tm4 = sim_timing ? now_us() : 0.0;
  rc = step_range(e, b, dst_lane0, (int)n_dst, 0, T_eff, 1, &oo, "gpf_simulate_batch");
  if (sim_timing)
    fprintf(stderr, "[gridpf] simulate_batch %lld lanes: gather + sync %.0f us, candidate rows %.0f, gpf_set_topology %.0f, prepare + mirror %.0f, plan + launch %.0f\n",
            n_dst, tm1 - tm0, tm2 - tm1, tm3 - tm2, tm4 - tm3, now_us() - tm4);
  return rc;
}

int gpf_step(gpf_handle e, int32_t t, int32_t max_iter, double tol_mva, double rebalance, int32_t cascade, float hard_overflow,
             float soft_overflow, int32_t nb_ts_allowed, int32_t max_rounds, int32_t is_dc) {
  gpf_step_opts o{};
  o.max_iter = max_iter; o.tol_mva = tol_mva; o.rebalance = rebalance; o.cascade = cascade; o.hard_overflow = hard_overflow;
  o.soft_overflow = soft_overflow; o.nb_ts_allowed = nb_ts_allowed; o.max_rounds = max_rounds; o.is_dc = is_dc; o.auto_reset = 0; o.warm_start = 0; o.track_cooldown = 0; o.nb_ts_reco = 0;
  return gpf_step_n(e, t, 1, &o);
}

int gpf_set_lane_redispatch(gpf_handle e, const float* delta_mw) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_lane_redispatch: null");
  HIP_TRY(hipSetDevice(e->device));
  if (!delta_mw) { e->has_delta = false; return GPF_OK; }
  const size_t n = (size_t)e->cap_lanes * e->g.n_gen;
  if (!e->lane_gen_delta.p) { HIP_TRY(e->lane_gen_delta.alloc(n)); HIP_TRY(hipMemsetAsync(e->lane_gen_delta.p, 0, n * sizeof(float), e->stream)); }
  HIP_TRY(hipMemcpyAsync(e->lane_gen_delta.p, delta_mw, (size_t)e->n_lanes * e->g.n_gen * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->has_delta = true;
  return GPF_OK;
}

int gpf_set_gen_limits(gpf_handle e, const double* pmin, const double* pmax, const double* ramp_up, const double* ramp_down,
                       const uint8_t* redispatchable, double eps_poly) {
  if (!e || !pmin || !pmax || !ramp_up || !ramp_down || !redispatchable) return fail(GPF_E_INVALID, "gpf_set_gen_limits: null");
  if (e->g.n_gen > gpf::RD_PER_LANE * gpf::WAVE) return fail(GPF_E_CAPACITY, "gpf_set_gen_limits: more than 256 generators");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const size_t ng = e->g.n_gen;
  e->rd_pmin.release(); e->rd_pmax.release(); e->rd_ru.release(); e->rd_rd.release(); e->rd_redisp.release();
  HIP_TRY(e->rd_pmin.upload(pmin, ng)); HIP_TRY(e->rd_pmax.upload(pmax, ng)); HIP_TRY(e->rd_ru.upload(ramp_up, ng));
  HIP_TRY(e->rd_rd.upload(ramp_down, ng)); HIP_TRY(e->rd_redisp.upload(redispatchable, ng));
  e->rd_eps = eps_poly;
  e->rd_ready = true;
  return GPF_OK;
}

int gpf_redispatch(gpf_handle e, int32_t lane0, int32_t n, const double* new_p, const double* prev_p, const double* actual,
                   const double* target, const uint8_t* modified, const double* rhs, int32_t apply, uint8_t* ok, float* actual_after) {
  if (!check_range(e, lane0, n) || !new_p || !prev_p || !actual || !target || !modified || !rhs)
    return fail(GPF_E_INVALID, "gpf_redispatch: bad arguments");
  if (!e->rd_ready) return fail(GPF_E_INVALID, "gpf_redispatch: call gpf_set_gen_limits first");
  if (n == 0) return GPF_OK;
  HIP_TRY(hipSetDevice(e->device));
  const size_t ng = e->g.n_gen, row = (size_t)n * ng;
  if (e->rd_in.n < 4 * row + (size_t)n) { e->rd_in.release(); HIP_TRY(e->rd_in.alloc(4 * row + n)); }
  if (e->rd_u8.n < row + (size_t)n) { e->rd_u8.release(); HIP_TRY(e->rd_u8.alloc(row + n)); }
  if (e->rd_after.n < row) { e->rd_after.release(); HIP_TRY(e->rd_after.alloc(row)); }
  double* d = e->rd_in.p;
  HIP_TRY(hipMemcpyAsync(d, new_p, row * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(d + row, prev_p, row * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(d + 2 * row, actual, row * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(d + 3 * row, target, row * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(d + 4 * row, rhs, (size_t)n * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->rd_u8.p, modified, row, hipMemcpyHostToDevice, e->stream));
  float* delta = nullptr;
  if (apply) {
    const size_t nd = (size_t)e->cap_lanes * ng;
    if (!e->lane_gen_delta.p) { HIP_TRY(e->lane_gen_delta.alloc(nd)); HIP_TRY(hipMemsetAsync(e->lane_gen_delta.p, 0, nd * sizeof(float), e->stream)); }
    delta = e->lane_gen_delta.p + (size_t)lane0 * ng;
    e->has_delta = true;
  }
  gpf::RedispDev R{};
  R.n_gen = (int)ng; R.eps_poly = e->rd_eps; R.pmin = e->rd_pmin.p; R.pmax = e->rd_pmax.p; R.ramp_up = e->rd_ru.p; R.ramp_down = e->rd_rd.p;
  R.redispatchable = e->rd_redisp.p;
  hipLaunchKernelGGL(gpf::redispatch_kernel, dim3(n), dim3(gpf::WAVE), 0, e->stream, R, n, d, d + row, d + 2 * row, d + 3 * row, e->rd_u8.p,
                     d + 4 * row, e->rd_u8.p + row, e->rd_after.p, delta);
  HIP_TRY(hipGetLastError());
  if (ok) HIP_TRY(hipMemcpyAsync(ok, e->rd_u8.p + row, (size_t)n, hipMemcpyDeviceToHost, e->stream));
  if (actual_after) HIP_TRY(hipMemcpyAsync(actual_after, e->rd_after.p, row * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_set_trajectory(gpf_handle e, int32_t n_steps_cap, int32_t what) {
  if (!e || n_steps_cap < 0 || (what & ~(GPF_TRAJ_RHO | GPF_TRAJ_OBS))) return fail(GPF_E_INVALID, "gpf_set_trajectory: bad arguments");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->traj_rho.release(); e->traj_status.release(); e->traj_cool.release();
  e->traj_out.release(); e->traj_topo.release(); e->traj_shb.release(); e->traj_lstat.release();
  e->traj_cap = 0; e->traj_what = 0; e->traj_valid = 0;
  if (n_steps_cap > 0 && what) {
    const size_t rows = (size_t)n_steps_cap * e->cap_lanes;
    const gpf::GridDev& g = e->g;
    HIP_TRY(e->traj_rho.alloc(rows * g.n_line));
    HIP_TRY(e->traj_status.alloc(rows));
    HIP_TRY(hipMemset(e->traj_status.p, 0xFF, rows));
    HIP_TRY(e->traj_cool.alloc(rows * g.n_line));                 // line cooldowns of every step (written when the step tracks them)
    HIP_TRY(hipMemset(e->traj_cool.p, 0, rows * g.n_line * sizeof(short)));
    if (what & GPF_TRAJ_OBS) {
      HIP_TRY(e->traj_out.alloc(rows * g.n_out)); HIP_TRY(e->traj_topo.alloc(rows * g.dim_topo));
      HIP_TRY(e->traj_shb.alloc(rows * std::max(g.n_shunt, 1))); HIP_TRY(e->traj_lstat.alloc(rows * g.n_line));
    }
    e->traj_cap = n_steps_cap;
    e->traj_what = what | GPF_TRAJ_RHO;
  }
  return GPF_OK;
}

}  // extern "C"
namespace {
// rows [step0, step0 + n_steps) x lanes [lane0, lane0 + n) of a [cap][cap_lanes][stride] device buffer -> dense host array
template <class T>
int traj_copy(gpf_engine* e, T* dst, const T* src, size_t stride, int step0, int n_steps, int lane0, int n) {
  if (!dst || stride == 0 || n == 0 || n_steps == 0) return GPF_OK;
  const size_t B = e->cap_lanes;
  HIP_TRY(hipMemcpy2DAsync(dst, (size_t)n * stride * sizeof(T), src + ((size_t)step0 * B + lane0) * stride, B * stride * sizeof(T),
                           (size_t)n * stride * sizeof(T), (size_t)n_steps, hipMemcpyDeviceToHost, e->stream));
  return GPF_OK;
}
}  // namespace
extern "C" {

int gpf_get_trajectory(gpf_handle e, int32_t step0, int32_t n_steps, int32_t lane0, int32_t n, float* rho, int8_t* status) {
  if (!check_range(e, lane0, n) || step0 < 0 || n_steps < 0 || step0 + n_steps > e->traj_valid)
    return fail(GPF_E_INVALID, "gpf_get_trajectory: bad range (only the steps of the last gpf_step_n are retrievable)");
  HIP_TRY(hipSetDevice(e->device));
  int rc = traj_copy(e, rho, e->traj_rho.p, (size_t)e->g.n_line, step0, n_steps, lane0, n);
  if (rc == GPF_OK) rc = traj_copy(e, reinterpret_cast<signed char*>(status), e->traj_status.p, 1, step0, n_steps, lane0, n);
  if (rc != GPF_OK) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_trajectory_cooldown(gpf_handle e, int32_t step0, int32_t n_steps, int32_t lane0, int32_t n, int16_t* line_cooldown) {
  if (!check_range(e, lane0, n) || !line_cooldown || step0 < 0 || n_steps < 0 || step0 + n_steps > e->traj_valid || !e->traj_cool.p)
    return fail(GPF_E_INVALID, "gpf_get_trajectory_cooldown: bad range (only the steps of the last gpf_step_n are retrievable)");
  HIP_TRY(hipSetDevice(e->device));
  int rc = traj_copy(e, line_cooldown, e->traj_cool.p, (size_t)e->g.n_line, step0, n_steps, lane0, n);
  if (rc != GPF_OK) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_cooldown(gpf_handle e, int32_t lane0, int32_t n, int32_t* line_cooldown) {
  if (!check_range(e, lane0, n) || !line_cooldown) return fail(GPF_E_INVALID, "gpf_get_cooldown: bad arguments");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(line_cooldown, e->cooldown.p + (size_t)lane0 * e->g.n_line, (size_t)n * e->g.n_line * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_set_cooldown(gpf_handle e, int32_t lane0, int32_t n, const int32_t* line_cooldown) {
  if (!check_range(e, lane0, n) || !line_cooldown) return fail(GPF_E_INVALID, "gpf_set_cooldown: bad arguments");
  for (size_t i = 0; i < (size_t)n * e->g.n_line; ++i) if (line_cooldown[i] < 0) return fail(GPF_E_INVALID, "gpf_set_cooldown: negative counter");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(e->cooldown.p + (size_t)lane0 * e->g.n_line, line_cooldown, (size_t)n * e->g.n_line * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_trajectory_obs(gpf_handle e, int32_t step0, int32_t n_steps, int32_t lane0, int32_t n, float* out, int32_t* topo_vect,
                           int32_t* shunt_bus, uint8_t* line_status) {
  if (!check_range(e, lane0, n) || step0 < 0 || n_steps < 0 || step0 + n_steps > e->traj_valid)
    return fail(GPF_E_INVALID, "gpf_get_trajectory_obs: bad range (only the steps of the last gpf_step_n are retrievable)");
  if (!(e->traj_what & GPF_TRAJ_OBS))
    return fail(GPF_E_INVALID, "gpf_get_trajectory_obs: no observation trajectory (gpf_set_trajectory(.., GPF_TRAJ_OBS))");
  HIP_TRY(hipSetDevice(e->device));
  const gpf::GridDev& g = e->g;
  int rc = traj_copy(e, out, e->traj_out.p, (size_t)g.n_out, step0, n_steps, lane0, n);
  if (rc == GPF_OK) rc = traj_copy(e, topo_vect, e->traj_topo.p, (size_t)g.dim_topo, step0, n_steps, lane0, n);
  if (rc == GPF_OK) rc = traj_copy(e, shunt_bus, e->traj_shb.p, (size_t)g.n_shunt, step0, n_steps, lane0, n);
  if (rc == GPF_OK) rc = traj_copy(e, line_status, e->traj_lstat.p, (size_t)g.n_line, step0, n_steps, lane0, n);
  if (rc != GPF_OK) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_episode(gpf_handle e, int32_t lane0, int32_t n, uint8_t* done, int32_t* steps_and_resets) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_get_episode: bad range");
  HIP_TRY(hipSetDevice(e->device));
  if (done) HIP_TRY(hipMemcpyAsync(done, e->done.p + lane0, (size_t)n, hipMemcpyDeviceToHost, e->stream));
  if (steps_and_resets)
    HIP_TRY(hipMemcpyAsync(steps_and_resets, e->episode.p + (size_t)lane0 * 2, (size_t)n * 2 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_set_overflow_count(gpf_handle e, int32_t lane0, int32_t n, const int32_t* overflow_count) {
  if (!check_range(e, lane0, n) || !overflow_count) return fail(GPF_E_INVALID, "gpf_set_overflow_count: bad arguments");
  for (size_t i = 0; i < (size_t)n * e->g.n_line; ++i) if (overflow_count[i] < 0) return fail(GPF_E_INVALID, "gpf_set_overflow_count: negative counter");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(e->overflow_count.p + (size_t)lane0 * e->g.n_line, overflow_count, (size_t)n * e->g.n_line * sizeof(int),
                         hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_step_outputs(gpf_handle e, int32_t lane0, int32_t n, float* rho, int32_t* overflow_count, int32_t* disc_round) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_get_step_outputs: bad range");
  HIP_TRY(hipSetDevice(e->device));
  const size_t nl = e->g.n_line;
  if (rho) HIP_TRY(hipMemcpyAsync(rho, e->rho.p + lane0 * nl, n * nl * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (overflow_count)
    HIP_TRY(hipMemcpyAsync(overflow_count, e->overflow_count.p + lane0 * nl, n * nl * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  if (disc_round)
    HIP_TRY(hipMemcpyAsync(disc_round, e->disc_round.p + lane0 * nl, n * nl * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_sync(gpf_handle e) {
  if (!e) return fail(GPF_E_INVALID, "gpf_sync: null");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(sync_stream_spin(e->stream));           // (polls for up to GRIDPF_SYNC_SPIN_US before blocking, see sync_stream_spin)
  return GPF_OK;
}

static int close_window(gpf_engine* e) {
  if (!e->window) return GPF_OK;
  if (e->win_launches > 0) {
    if (!e->win_marked) HIP_TRY(hipEventRecord(e->win_b, e->stream));     // (mode 3 already recorded it behind the last launch)
    HIP_TRY(hipEventSynchronize(e->win_b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->win_a, e->win_b));
    e->acc_ms += ms;
    e->acc_launches += e->win_launches;
  }
  e->win_launches = 0;
  e->window = false;
  e->win_marked = false;
  return GPF_OK;
}

static int open_window(gpf_engine* e) {
  if (!e->win_a) { HIP_TRY(hipEventCreate(&e->win_a)); HIP_TRY(hipEventCreate(&e->win_b)); }
  HIP_TRY(hipEventRecord(e->win_a, e->stream));
  e->win_launches = 0;
  e->window = true;
  e->win_marked = false;
  return GPF_OK;
}

int gpf_set_profiling(gpf_handle e, int32_t mode) {
  if (!e) return fail(GPF_E_INVALID, "gpf_set_profiling: null");
  HIP_TRY(hipSetDevice(e->device));
  if (mode == 3) {                      // end of the running window = this point of the stream (asynchronous; read by the next call)
    if (e->window && e->win_launches > 0) { HIP_TRY(hipEventRecord(e->win_b, e->stream)); e->win_marked = true; }
    return GPF_OK;
  }
  int rc = close_window(e);
  if (rc != GPF_OK) return rc;
  e->profiling = mode == 2;
  if (mode == 1) return open_window(e);
  return GPF_OK;
}

int gpf_get_kernel_time(gpf_handle e, double* total_ms, int64_t* n_launches) {
  if (!e) return fail(GPF_E_INVALID, "gpf_get_kernel_time: null");
  HIP_TRY(hipSetDevice(e->device));
  int rc = drain_events(e);
  if (rc != GPF_OK) return rc;
  if (e->window) {                      // close the running window and open the next one
    rc = close_window(e);
    if (rc == GPF_OK) rc = open_window(e);
    if (rc != GPF_OK) return rc;
  }
  if (total_ms) *total_ms = e->acc_ms;
  if (n_launches) *n_launches = e->acc_launches;
  e->acc_ms = 0.0;
  e->acc_launches = 0;
  return GPF_OK;
}


/* ---- DC sensitivity (PTDF) path ------------------------------------------------------------------------------------ */
int gpf_ptdf_build(gpf_handle e, int32_t lane) {
  if (!check_range(e, lane, 1)) return fail(GPF_E_INVALID, "gpf_ptdf_build: bad lane");
  HIP_TRY(hipSetDevice(e->device));
  const gpf::GridDev& g = e->g;
  const gpf::OutOff& oo = e->oo;
  std::vector<int> topo(g.dim_topo), sb(std::max(g.n_shunt, 1));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(topo.data(), e->topo.p + (size_t)lane * g.dim_topo, (size_t)g.dim_topo * sizeof(int), hipMemcpyDeviceToHost));
  if (g.n_shunt) HIP_TRY(hipMemcpy(sb.data(), e->shunt_bus.p + (size_t)lane * g.n_shunt, (size_t)g.n_shunt * sizeof(int), hipMemcpyDeviceToHost));
  const int nbt = g.nb_tot;
  auto bus_of = [&](int sub, int local) -> int { return (local >= 1 && local <= g.n_busbar) ? sub + (local - 1) * g.n_sub : -1; };
  std::vector<char> act(nbt, 0), ref(nbt, 0);
  std::vector<int> lf(g.n_line, -1), lt(g.n_line, -1);
  for (int l = 0; l < g.n_line; ++l) {
    const int bo = topo[e->h_line_or_pos[l]], be = topo[e->h_line_ex_pos[l]];
    if (bo >= 1 && be >= 1) {
      lf[l] = bus_of(e->h_line_or_sub[l], bo); lt[l] = bus_of(e->h_line_ex_sub[l], be);
      if (lf[l] < 0 || lt[l] < 0) return fail(GPF_E_INVALID, "gpf_ptdf_build: bus id out of range");
      act[lf[l]] = act[lt[l]] = 1;
    }
  }
  std::vector<int> inj_bus(g.n_inj, -1);
  std::vector<double> inj_w(g.n_inj, 0.0);
  for (int i = 0; i < g.n_gen; ++i) {
    const int b = bus_of(e->h_gen_sub[i], topo[e->h_gen_pos[i]]);
    if (b < 0) continue;
    act[b] = 1;
    if (e->h_gen_slack[i]) ref[b] = 1; else { inj_bus[oo.inj_gen_p + i] = b; inj_w[oo.inj_gen_p + i] = 1.0; }
  }
  for (int i = 0; i < g.n_load; ++i) {
    const int b = bus_of(e->h_load_sub[i], topo[e->h_load_pos[i]]);
    if (b >= 0) { act[b] = 1; inj_bus[oo.inj_load_p + i] = b; inj_w[oo.inj_load_p + i] = -1.0; }
  }
  for (int i = 0; i < g.n_sto; ++i) {
    const int b = bus_of(e->h_sto_sub[i], topo[e->h_sto_pos[i]]);
    if (b >= 0) { act[b] = 1; inj_bus[oo.inj_sto_p + i] = b; inj_w[oo.inj_sto_p + i] = -1.0; }
  }
  for (int i = 0; i < g.n_shunt; ++i) {
    const int b = bus_of(e->h_shunt_sub[i], sb[i]);
    if (b >= 0) { act[b] = 1; inj_bus[oo.inj_sh_p + i] = b; inj_w[oo.inj_sh_p + i] = -e->h_shunt_fact[i]; }
  }
  // reduced B' over the active non-reference buses, inverted by Gauss-Jordan with partial pivoting (once per topology)
  std::vector<int> idx(nbt, -1), buses;
  bool any_ref = false;
  for (int b = 0; b < nbt; ++b) { any_ref |= (act[b] && ref[b]); if (act[b] && !ref[b]) { idx[b] = (int)buses.size(); buses.push_back(b); } }
  if (!any_ref) return fail(GPF_E_INVALID, "gpf_ptdf_build: no in-service slack generator in this topology");
  const int nr = (int)buses.size();
  std::vector<double> M((size_t)nr * 2 * nr, 0.0);
  for (int r = 0; r < nr; ++r) M[(size_t)r * 2 * nr + nr + r] = 1.0;
  for (int l = 0; l < g.n_line; ++l) {
    if (lf[l] < 0 || lf[l] == lt[l]) continue;
    const double bb = e->h_br_bdc[l];
    const int a = idx[lf[l]], c = idx[lt[l]];
    if (a >= 0) M[(size_t)a * 2 * nr + a] += bb;
    if (c >= 0) M[(size_t)c * 2 * nr + c] += bb;
    if (a >= 0 && c >= 0) { M[(size_t)a * 2 * nr + c] -= bb; M[(size_t)c * 2 * nr + a] -= bb; }
  }
  for (int k = 0; k < nr; ++k) {
    int p = k;
    for (int r = k + 1; r < nr; ++r) if (std::fabs(M[(size_t)r * 2 * nr + k]) > std::fabs(M[(size_t)p * 2 * nr + k])) p = r;
    const double pv = M[(size_t)p * 2 * nr + k];
    if (!(std::fabs(pv) > 1e-12)) return fail(GPF_E_INVALID, "gpf_ptdf_build: the topology is islanded (singular B')");
    if (p != k) for (int q = 0; q < 2 * nr; ++q) std::swap(M[(size_t)k * 2 * nr + q], M[(size_t)p * 2 * nr + q]);
    const double rp = 1.0 / pv;
    for (int q = 0; q < 2 * nr; ++q) M[(size_t)k * 2 * nr + q] *= rp;
    for (int r = 0; r < nr; ++r) {
      if (r == k) continue;
      const double mlt = M[(size_t)r * 2 * nr + k];
      if (mlt == 0.0) continue;
      for (int q = k; q < 2 * nr; ++q) M[(size_t)r * 2 * nr + q] -= mlt * M[(size_t)k * 2 * nr + q];
    }
  }
  auto X = [&](int bus_row, int bus_col) -> double {
    const int r = bus_row >= 0 ? idx[bus_row] : -1, c = idx[bus_col];
    return (r >= 0 && c >= 0) ? M[(size_t)r * 2 * nr + nr + c] : 0.0;
  };
  e->h_ptdf.assign((size_t)g.n_line * nbt, 0.0);
  // the device GEMM runs over the ACTIVE buses only (compact index: half of the n_sub * n_busbar ids are unused)
  std::vector<int> compact(nbt, -1);
  int n_act = 0;
  for (int b = 0; b < nbt; ++b) if (act[b]) compact[b] = n_act++;
  const int nb_pad = std::max(4, (n_act + 3) & ~3), line_pad = (g.n_line + 15) & ~15;
  const int kpad = (nb_pad + 31) & ~31;                      // (rows behind nb_pad stay zero: gpf_ptdf_flows_rows runs whole trips of 8 k-steps)
  std::vector<double> pt((size_t)kpad * line_pad, 0.0);
  for (int l = 0; l < g.n_line; ++l) {
    if (lf[l] < 0 || lf[l] == lt[l]) continue;
    for (int b = 0; b < nbt; ++b) {
      if (idx[b] < 0) continue;
      const double v = e->h_br_bdc[l] * (X(lf[l], b) - X(lt[l], b));
      e->h_ptdf[(size_t)l * nbt + b] = v;
      pt[(size_t)compact[b] * line_pad + l] = v;
    }
  }
  for (int i = 0; i < g.n_inj; ++i) if (inj_bus[i] >= 0) inj_bus[i] = compact[inj_bus[i]];
  e->ptdf_ready = false;
  e->ptdf_inj_bus.release(); e->ptdf_inj_w.release(); e->ptdf_t.release();
  HIP_TRY(e->ptdf_inj_bus.upload(inj_bus.data(), inj_bus.size()));
  HIP_TRY(e->ptdf_inj_w.upload(inj_w.data(), inj_w.size()));
  HIP_TRY(e->ptdf_t.upload(pt.data(), pt.size()));
  if (e->ptdf_nb_pad != nb_pad || e->ptdf_line_pad != line_pad || !e->ptdf_flow.p) {
    e->ptdf_flow.release();
    HIP_TRY(e->ptdf_flow.alloc((size_t)e->cap_lanes * line_pad));
  }
  {   // LODF[l][k] = H[l][k] / (1 - H[k][k]), H[l][k] = PTDF[l][from_k] - PTDF[l][to_k]; LODF[k][k] = -1
    std::vector<double> lo_((size_t)g.n_line * line_pad, 0.0);
    // elements on every bus: a line whose outage only removes a bus that carries nothing else (one line end, no injection) does not
    // island anything -- the reference's DC power flow of that contingency converges with every other flow unchanged: column of zeros
    std::vector<int> n_lines_at(nbt, 0), n_other_at(nbt, 0);
    for (int l = 0; l < g.n_line; ++l) if (lf[l] >= 0) { ++n_lines_at[lf[l]]; ++n_lines_at[lt[l]]; }
    for (int i = 0; i < g.n_gen; ++i) { const int b = bus_of(e->h_gen_sub[i], topo[e->h_gen_pos[i]]); if (b >= 0) ++n_other_at[b]; }
    for (int i = 0; i < g.n_load; ++i) { const int b = bus_of(e->h_load_sub[i], topo[e->h_load_pos[i]]); if (b >= 0) ++n_other_at[b]; }
    for (int i = 0; i < g.n_sto; ++i) { const int b = bus_of(e->h_sto_sub[i], topo[e->h_sto_pos[i]]); if (b >= 0) ++n_other_at[b]; }
    for (int i = 0; i < g.n_shunt; ++i) { const int b = bus_of(e->h_shunt_sub[i], sb[i]); if (b >= 0) ++n_other_at[b]; }
    for (int k = 0; k < g.n_line; ++k) {
      if (lf[k] < 0 || lf[k] == lt[k]) continue;                 // an open line: its outage changes nothing
      const double hkk = e->h_ptdf[(size_t)k * nbt + lf[k]] - e->h_ptdf[(size_t)k * nbt + lt[k]];
      const double den = 1.0 - hkk;
      const bool dangling = (n_lines_at[lf[k]] == 1 && n_other_at[lf[k]] == 0) || (n_lines_at[lt[k]] == 1 && n_other_at[lt[k]] == 0);
      for (int l = 0; l < g.n_line; ++l) {
        const double hlk = e->h_ptdf[(size_t)l * nbt + lf[k]] - e->h_ptdf[(size_t)l * nbt + lt[k]];
        lo_[(size_t)l * line_pad + k] = std::fabs(den) < 1e-8 ? (dangling ? (l == k ? -1.0 : 0.0) : std::nan("")) : (l == k ? -1.0 : hlk / den);   // (diagonal -1 as in the batch builder)
      }
    }
    e->lodf.release();
    { std::vector<float> lof(lo_.begin(), lo_.end()); HIP_TRY(e->lodf.upload(lof.data(), lof.size())); }
    e->lodf_worst.release();
    HIP_TRY(e->lodf_worst.alloc((size_t)e->cap_lanes * line_pad));
  }
  e->ptdf_nb_pad = nb_pad; e->ptdf_line_pad = line_pad;
  e->ptdf_ready = true;
  e->ptdf_batch = false;
  return GPF_OK;
}

// PtdfDev of the tables the flows / screening kernels run on: the single topology of gpf_ptdf_build or the class tables of gpf_ptdf_build_batch
static gpf::PtdfDev ptdf_dev(gpf_engine* e) {
  gpf::PtdfDev P{};
  P.n_inj = e->g.n_inj; P.nb_pad = e->ptdf_nb_pad; P.line_pad = e->ptdf_line_pad; P.n_line = e->g.n_line;
  if (e->ptdf_batch) {
    P.inj_bus = nullptr; P.inj_w = e->ptdfb_inj_w.p; P.ptdf_t = e->ptdfb_t.p;
    P.order = e->ptdfb_order.p; P.blk_class = e->ptdfb_blk_class.p; P.cls_desc = e->ptdfb_desc.p; P.cls_status = e->ptdfb_status.p;
    P.desc_stride = e->ptdfb_desc_stride; P.inj_bus_off = gpf::PTDFB_HDR + 2 * e->g.n_line;
    P.ptdf_stride = (long long)e->ptdfb_kpad * e->ptdf_line_pad;
  } else {
    P.inj_bus = e->ptdf_inj_bus.p; P.inj_w = e->ptdf_inj_w.p; P.ptdf_t = e->ptdf_t.p;
  }
  return P;
}

// A few PERSISTENT host threads for the table walks of gpf_ptdf_build_batch (row hashes / comparisons, descriptors of unseen classes): creating
// threads per call cost more than the work it spread (measured on the MI355X box: no gain from 4 fresh std::threads on 0.8 ms of work).
// Workers sleep on a condition variable between calls; they are detached at process exit (never joined: no ordering against the HIP runtime).
namespace {
class HostPool {
 public:
  static HostPool& get() { static HostPool* p = new HostPool(); return *p; }      // (intentionally leaked)
  int size() const { return n_workers_ + 1; }
  // body(part, n_parts) for part = 0 .. n_parts - 1, part 0 on the caller's thread; returns when all parts are done
  void run(int n_parts, const std::function<void(int, int)>& body) {
    n_parts = std::max(1, std::min(n_parts, size()));
    if (n_parts == 1) { body(0, 1); return; }
    std::lock_guard<std::mutex> call_lk(call_mu_);          // one parallel region at a time
    {
      std::lock_guard<std::mutex> lk(mu_);
      body_ = &body; parts_ = n_parts; next_ = 1; left_ = n_parts - 1; ++gen_;
    }
    cv_.notify_all();
    body(0, n_parts);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return left_ == 0; });
    body_ = nullptr;
  }
 private:
  HostPool() {
    const char* v = getenv("GRIDPF_PTDFB_THREADS");
    int t = v ? atoi(v) : 4;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && t > hw) t = hw;
    n_workers_ = std::max(0, t - 1);
    for (int w = 0; w < n_workers_; ++w) std::thread([this] { loop(); }).detach();
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return gen_ != seen && next_ < parts_; });
      const unsigned long long g = gen_;
      while (gen_ == g && next_ < parts_) {
        const int part = next_++;
        const std::function<void(int, int)>* b = body_;
        const int np = parts_;
        lk.unlock();
        (*b)(part, np);
        lk.lock();
        if (--left_ == 0) done_.notify_all();
      }
      seen = g;
    }
  }
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int, int)>* body_ = nullptr;
  int n_workers_ = 0, parts_ = 0, next_ = 0, left_ = 0;
  unsigned long long gen_ = 0;
};
}  // namespace

// completes the asynchronous tail of gpf_ptdf_build_batch: class status in h_ptdfb_status, kernel duration in ptdfb_kernel_ms
static int ptdfb_finish(gpf_engine* e) {
  if (!e->ptdfb_pending) return GPF_OK;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::copy(e->ptdfb_status_pin, e->ptdfb_status_pin + e->h_ptdfb_status.size(), e->h_ptdfb_status.begin());
  if (e->ptdfb_prefetched && e->ptdfb_host_stale) {            // (device path: the class map and the descriptor headers came back behind the status)
    const int n = e->ptdfb_n, nc = e->ptdfb_classes;
    e->h_ptdfb_lane_class.assign(e->ptdfg_back_pin, e->ptdfg_back_pin + n);
    e->h_ptdfb_hdr.assign(e->ptdfg_back_pin + n, e->ptdfg_back_pin + n + (size_t)nc * 4);
    e->ptdfb_host_stale = false;
  }
  e->ptdfb_prefetched = false;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e->ptdfb_ev_a, e->ptdfb_ev_b);
  e->ptdfb_kernel_ms = ms;
  e->ptdfb_pending = false;
  return GPF_OK;
}

// lane -> class map and descriptor headers of a build whose integer half ran on the device: to the host mirrors, on demand (12 KB for 2 048
// lanes / 256 classes; the descriptors themselves stay on the device); with_bus: the compact -> bus maps too (gpf_ptdf_batch_get)
static int ptdfb_fetch_host(gpf_engine* e, bool with_bus = false) {
  if (!e->ptdfb_host_stale && !(with_bus && e->ptdfb_bus_stale)) return GPF_OK;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int n = e->ptdfb_n, nc = e->ptdfb_classes, stride = e->ptdfb_desc_stride, nbt = e->g.nb_tot;
  if (e->ptdfb_host_stale) {
    e->h_ptdfb_lane_class.resize(n);
    HIP_TRY(hipMemcpy(e->h_ptdfb_lane_class.data(), e->ptdfg_lane_class.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    e->h_ptdfb_hdr.resize((size_t)nc * 4);
    HIP_TRY(hipMemcpy2D(e->h_ptdfb_hdr.data(), 4 * sizeof(int), e->ptdfb_desc.p, (size_t)stride * sizeof(int), 4 * sizeof(int), (size_t)nc, hipMemcpyDeviceToHost));
    e->ptdfb_host_stale = false;
  }
  if (with_bus && e->ptdfb_bus_stale) {
    std::vector<int> c2b((size_t)nc * nbt);
    HIP_TRY(hipMemcpy(c2b.data(), e->ptdfg_c2b.p, c2b.size() * sizeof(int), hipMemcpyDeviceToHost));
    e->h_ptdfb_bus.assign(nc, std::vector<int>());
    for (int c = 0; c < nc; ++c) {
      const int n_act = e->h_ptdfb_hdr[(size_t)c * 4 + 1];
      e->h_ptdfb_bus[c].assign(c2b.begin() + (size_t)c * nbt, c2b.begin() + (size_t)c * nbt + n_act);
    }
    e->ptdfb_bus_stale = false;
  }
  return GPF_OK;
}

// The integer half of gpf_ptdf_build_batch on the device.  out[0..3] = classes, slots, largest n_pad, largest n_act.  Returns GPF_OK with
// *done = false when the device path does not apply or flagged something (hash collision inside a class, bus id out of range, capacity):
// the caller then takes the host path, which reports the error properly.
static int ptdfb_group_on_device(gpf_engine* e, int lane0, int n, int stride, int out[4], bool* done) {
  *done = false;
  const gpf::GridDev& g = e->g;
  if (n > gpf::PTDFG_MAX_LANES || g.nb_tot > gpf::PTDFG_MAX_BUS || g.n_line > 256 || getenv("GRIDPF_PTDFB_HOST")) return GPF_OK;
  HIP_TRY(e->ptdfg_hash.ensure((size_t)n)); HIP_TRY(e->ptdfg_lane_class.ensure((size_t)n)); HIP_TRY(e->ptdfg_first.ensure((size_t)n));
  HIP_TRY(e->ptdfb_order.ensure((size_t)16 * n)); HIP_TRY(e->ptdfb_blk_class.ensure((size_t)n));
  HIP_TRY(e->ptdfb_desc.ensure((size_t)n * stride)); HIP_TRY(e->ptdfg_c2b.ensure((size_t)n * g.nb_tot)); HIP_TRY(e->ptdfg_info.ensure(8));
  if (!e->ptdfg_info_pin) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->ptdfg_info_pin), 8 * sizeof(int), hipHostMallocDefault));
  gpf::PtdfGroupDev D{};
  D.topo = e->topo.p; D.shunt_bus = e->shunt_bus.p;
  D.lane0 = lane0; D.n = n; D.dim_topo = g.dim_topo; D.n_shunt = g.n_shunt; D.n_sub = g.n_sub; D.n_busbar = g.n_busbar; D.n_line = g.n_line;
  D.n_gen = g.n_gen; D.n_load = g.n_load; D.n_sto = g.n_sto; D.n_inj = g.n_inj;
  D.inj_gen_p = e->oo.inj_gen_p; D.inj_load_p = e->oo.inj_load_p; D.inj_sto_p = e->oo.inj_sto_p; D.inj_sh_p = e->oo.inj_sh_p;
  D.line_or_pos = e->line_or_pos.p; D.line_ex_pos = e->line_ex_pos.p; D.line_or_sub = e->line_or_sub.p; D.line_ex_sub = e->line_ex_sub.p;
  D.gen_pos = e->gen_pos.p; D.gen_sub = e->gen_sub.p; D.load_pos = e->load_pos.p; D.load_sub = e->load_sub.p; D.sto_pos = e->sto_pos.p;
  D.sto_sub = e->sto_sub.p; D.shunt_sub = e->shunt_sub.p; D.gen_slack = e->gen_slack.p;
  D.desc_stride = stride;
  D.hash = e->ptdfg_hash.p; D.lane_class = e->ptdfg_lane_class.p; D.first_lane = e->ptdfg_first.p; D.order = e->ptdfb_order.p;
  D.blk_class = e->ptdfb_blk_class.p; D.desc = e->ptdfb_desc.p; D.c2b = e->ptdfg_c2b.p; D.info = e->ptdfg_info.p;
  hipLaunchKernelGGL(gpf::ptdfg_hash_kernel, dim3(n), dim3(64), 0, e->stream, D);
  int np2 = gpf::PTDFG_SORT_THREADS;                   // (ptdfg_group_kernel sorts a multiple of its workgroup)
  while (np2 < n) np2 <<= 1;
  const size_t lds_sort = (size_t)np2 * 20;
  static size_t lds_sort_set[64] = {0};
  if (lds_sort > lds_sort_set[e->device & 63]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gpf::ptdfg_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
    lds_sort_set[e->device & 63] = lds_sort;
  }
  hipLaunchKernelGGL(gpf::ptdfg_group_kernel, dim3(1), dim3(gpf::PTDFG_SORT_THREADS), lds_sort, e->stream, D);
  hipLaunchKernelGGL(gpf::ptdfg_verify_kernel, dim3(n), dim3(64), 0, e->stream, D);
  hipLaunchKernelGGL(gpf::ptdfg_desc_kernel, dim3(n), dim3(gpf::PTDFG_DESC_THREADS), 0, e->stream, D);     // (blocks beyond the class count return at once)
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(e->ptdfg_info_pin, e->ptdfg_info.p, 8 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int* info = e->ptdfg_info_pin;
  if (info[4] || info[5] || info[0] <= 0) return GPF_OK;           // -> host path
  out[0] = info[0]; out[1] = info[1]; out[2] = std::max(16, info[2]); out[3] = std::max(1, info[3]);
  *done = true;
  return GPF_OK;
}

/* ---- PTDF / LODF of every distinct topology of a lane range, built on the device (gridpf_ptdf_batch.hpp) --------------------------- */
int gpf_ptdf_build_batch(gpf_handle e, int32_t lane0, int32_t n, int32_t with_lodf, int32_t* n_classes_out) {
  if (!check_range(e, lane0, n) || n <= 0) return fail(GPF_E_INVALID, "gpf_ptdf_build_batch: bad lane range");
  HIP_TRY(hipSetDevice(e->device));
  // a rebuild overwrites the lane -> class map and may regrow the device tables before it can fail: from here until it has succeeded there
  // are NO tables (gpf_ptdf_flows / gpf_ptdf_batch_get refuse), instead of new lane classes against old tables
  e->ptdf_ready = false; e->ptdf_batch = false;
  e->ptdfb_pending = false;                        // (a status nobody asked for: the stream orders the next build behind the last one)
  const gpf::GridDev& g = e->g;
  const gpf::OutOff& oo = e->oo;
  const int nl = g.n_line, nbt = g.nb_tot, nsh = g.n_shunt;
  static const bool stage_timing = getenv("GRIDPF_SIM_TIMING") != nullptr;      // developer: stage times of the call on stderr
  auto now_us = [] { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3; };
  double tm[6] = {0, 0, 0, 0, 0, 0};
  // ---- the integer half: on the device (gridpf_ptdf_group.hpp) when the range allows it, else -- and whenever the device flagged something -- on the host
  const int stride = (gpf::PTDFB_HDR + 3 * nl + g.n_inj + gpf::PTDFB_MAX_N + 1 + 2 * nl + 3) & ~3;
  int nc = 0, npad_max = 16, nact_max = 1;
  size_t n_slots = 0;
  std::vector<int> desc, order, blk_class;
  int grp[4] = {0, 0, 0, 0};
  bool dev = false;
  { const int rc_g = ptdfb_group_on_device(e, lane0, n, stride, grp, &dev); if (rc_g != GPF_OK) return rc_g; }
  auto host_group = [&]() -> int {
  // the lanes' topology rows as they are on the device (a cascade inside gpf_step_n may have tripped lines the host never saw)
  // (when no kernel can have rewritten them -- no cascade, no outage tables since the engine was created -- and the host sent every row of the
  //  range itself, the host mirrors ARE the device rows: no trip over PCIe, no synchronisation; 2 048 rows of 560 ints are 4.6 MB)
  std::vector<int> topo_own, sb_own;
  const int* topo_p = nullptr;
  const int* sb_p = nullptr;
  bool mirror_ok = !e->dev_topo_dirty && getenv("GRIDPF_PTDFB_NO_MIRROR") == nullptr;
  for (int k = lane0; k < lane0 + n && mirror_ok; ++k)
    mirror_ok = e->h_lane_topo[(size_t)k * g.dim_topo] != INT_MIN && (!nsh || e->h_lane_sb[(size_t)k * nsh] != INT_MIN);
  if (stage_timing) tm[0] = now_us();
  if (mirror_ok) {
    topo_p = e->h_lane_topo.data() + (size_t)lane0 * g.dim_topo;
    sb_p = e->h_lane_sb.data() + (size_t)lane0 * std::max(nsh, 1);
  } else {
    topo_own.resize((size_t)n * g.dim_topo); sb_own.resize((size_t)n * std::max(nsh, 1));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (stage_timing) tm[0] = now_us();
    HIP_TRY(hipMemcpy(topo_own.data(), e->topo.p + (size_t)lane0 * g.dim_topo, topo_own.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (nsh) HIP_TRY(hipMemcpy(sb_own.data(), e->shunt_bus.p + (size_t)lane0 * nsh, (size_t)n * nsh * sizeof(int), hipMemcpyDeviceToHost));
    topo_p = topo_own.data(); sb_p = sb_own.data();
  }
  struct RowView { const int* p; const int* data() const { return p; } int operator[](size_t i) const { return p[i]; } } topo{topo_p}, sb{sb_p};
  if (stage_timing) tm[1] = now_us();
  // ---- classes: lanes with identical (topology row, shunt buses) ----------------------------------------------------------------------
  std::unordered_map<uint64_t, std::vector<int>> cls_of;     // row hash -> classes with that hash (rows compared on a hit)
  std::vector<int> first_lane;                       // representative lane (index in the range) of each class
  std::vector<uint64_t> first_hash;                  // its row hash (key of the descriptor cache)
  e->h_ptdfb_lane_class.assign(n, -1);
  auto same_rows = [&](int a, int b) {
    return std::memcmp(topo.data() + (size_t)a * g.dim_topo, topo.data() + (size_t)b * g.dim_topo, (size_t)g.dim_topo * sizeof(int)) == 0 &&
           (!nsh || std::memcmp(sb.data() + (size_t)a * nsh, sb.data() + (size_t)b * nsh, (size_t)nsh * sizeof(int)) == 0);
  };
  cls_of.reserve((size_t)n);
  // host threads for the two table walks of this call (row hashes, descriptors of unseen classes): a few hundred microseconds of one core each
  // for 2 048 lanes / 256 classes of a 118-substation grid, embarrassingly parallel
  HostPool& pool = HostPool::get();
  auto par_for = [&](int count, int min_per_thread, const std::function<void(int, int, int)>& body) {     // body(begin, end, part index)
    const int nt = std::max(1, std::min(pool.size(), count / std::max(1, min_per_thread)));
    pool.run(nt, [&](int part, int n_parts) { body((int)((long long)count * part / n_parts), (int)((long long)count * (part + 1) / n_parts), part); });
  };
  std::vector<uint64_t> row_hash((size_t)n);
  par_for(n, 256, [&](int k0, int k1, int) {
  for (int k = k0; k < k1; ++k) {
    // row hash: four independent multiply-xor chains over the row (one dependent multiply per int was 1 ms for 2 048 rows of 560 ints);
    // equal hashes are confirmed by comparing the rows, so the hash only has to spread
    const int* tp = topo.data() + (size_t)k * g.dim_topo;
    uint64_t h0 = 1469598103934665603ull, h1 = 0x9E3779B97F4A7C15ull, h2 = 0xC2B2AE3D27D4EB4Full, h3 = 0x165667B19E3779F9ull;
    int i = 0;
    for (; i + 8 <= g.dim_topo; i += 8) {
      h0 = (h0 ^ ((uint64_t)(uint32_t)tp[i] | ((uint64_t)(uint32_t)tp[i + 1] << 32))) * 0x9FB21C651E98DF25ull;
      h1 = (h1 ^ ((uint64_t)(uint32_t)tp[i + 2] | ((uint64_t)(uint32_t)tp[i + 3] << 32))) * 0xD6E8FEB86659FD93ull;
      h2 = (h2 ^ ((uint64_t)(uint32_t)tp[i + 4] | ((uint64_t)(uint32_t)tp[i + 5] << 32))) * 0xA0761D6478BD642Full;
      h3 = (h3 ^ ((uint64_t)(uint32_t)tp[i + 6] | ((uint64_t)(uint32_t)tp[i + 7] << 32))) * 0xE7037ED1A0B428DBull;
    }
    for (; i < g.dim_topo; ++i) h0 = (h0 ^ (uint32_t)tp[i]) * 1099511628211ull;
    for (int q = 0; q < nsh; ++q) h1 = (h1 ^ ((uint32_t)sb[(size_t)k * nsh + q] + 0x9E3779B9u)) * 1099511628211ull;
    uint64_t h = h0 ^ (h1 >> 29 | h1 << 35) ^ (h2 >> 17 | h2 << 47) ^ (h3 >> 41 | h3 << 23);
    h ^= h >> 32;
    row_hash[k] = h;
  }
  });
  // classes by hash first (no row is touched), then every lane's row is compared with its class representative's -- in parallel: the two
  // passes over the rows (hash, confirm) are what grouping costs (4.6 MB each for 2 048 lanes of a 118-substation grid, memory-bound on one core)
  for (int k = 0; k < n; ++k) {
    const uint64_t h = row_hash[k];
    std::vector<int>& cand = cls_of[h];
    if (cand.empty()) { cand.push_back((int)first_lane.size()); first_lane.push_back(k); first_hash.push_back(h); }
    e->h_ptdfb_lane_class[k] = cand[0];
  }
  std::vector<char> differs((size_t)n, 0);
  par_for(n, 256, [&](int k0, int k1, int) {
    for (int k = k0; k < k1; ++k) { const int rep = first_lane[e->h_ptdfb_lane_class[k]]; differs[k] = (rep != k && !same_rows(rep, k)) ? 1 : 0; }
  });
  for (int k = 0; k < n; ++k) {                     // (a 64-bit hash collision between different rows: never seen; handled the slow way)
    if (!differs[k]) continue;
    const uint64_t h = row_hash[k];
    std::vector<int>& cand = cls_of[h];
    int c = -1;
    for (size_t q = 1; q < cand.size(); ++q) if (same_rows(first_lane[cand[q]], k)) { c = cand[q]; break; }
    if (c < 0) { c = (int)first_lane.size(); first_lane.push_back(k); first_hash.push_back(h); cand.push_back(c); }
    e->h_ptdfb_lane_class[k] = c;
  }
  nc = (int)first_lane.size();
  if (stage_timing) tm[2] = now_us();
  // descriptor: header | lf | lt | inj_bus | lflag | row pointers of B' [PTDFB_MAX_N + 1] | row entries [2 n_line] (gridpf_ptdf_batch.hpp)
  if (nl > 65535) return fail(GPF_E_CAPACITY, "gpf_ptdf_build_batch: more than 65535 lines");
  desc.assign((size_t)nc * stride, -1);
  e->h_ptdfb_bus.assign(nc, std::vector<int>());
  auto bus_of = [&](int sub, int local) -> int { return (local >= 1 && local <= g.n_busbar) ? sub + (local - 1) * g.n_sub : -1; };
  // One descriptor per class: ~3 us of table walks each (800 us for 256 classes of a 118-substation grid on one core)
  struct ClsScratch { std::vector<char> act, ref, has_ref; std::vector<int> bf, bt, n_lines_at, n_other_at, ibus, compact, comp, cnt; int npad_max = 16, nact_max = 1; };
  auto build_class = [&](int c, ClsScratch& S_) -> int {
    std::vector<char>&act = S_.act, &ref = S_.ref, &has_ref = S_.has_ref;
    std::vector<int>&bf = S_.bf, &bt = S_.bt, &n_lines_at = S_.n_lines_at, &n_other_at = S_.n_other_at, &ibus = S_.ibus, &compact = S_.compact, &comp = S_.comp, &cnt = S_.cnt;
    const int* tp = topo.data() + (size_t)first_lane[c] * g.dim_topo;
    const int* sbp = sb.data() + (size_t)first_lane[c] * std::max(nsh, 1);
    int* d = desc.data() + (size_t)c * stride;
    int* lf = d + gpf::PTDFB_HDR;
    int* lt = lf + nl;
    int* ib = lt + nl;
    int* lflag = ib + g.n_inj;
    act.assign(nbt, 0); ref.assign(nbt, 0);
    bf.assign(nl, -1); bt.assign(nl, -1);
    n_lines_at.assign(nbt, 0); n_other_at.assign(nbt, 0);      // in-service line ends / other elements on each bus
    for (int l = 0; l < nl; ++l) {
      const int bo = tp[e->h_line_or_pos[l]], be = tp[e->h_line_ex_pos[l]];
      if (bo >= 1 && be >= 1) {
        bf[l] = bus_of(e->h_line_or_sub[l], bo); bt[l] = bus_of(e->h_line_ex_sub[l], be);
        if (bf[l] < 0 || bt[l] < 0) return 1;
        act[bf[l]] = act[bt[l]] = 1;
      }
    }
    ibus.assign(g.n_inj, -1);
    for (int i = 0; i < g.n_gen; ++i) {
      const int b = bus_of(e->h_gen_sub[i], tp[e->h_gen_pos[i]]);
      if (b < 0) continue;
      act[b] = 1;
      ++n_other_at[b];
      if (e->h_gen_slack[i]) ref[b] = 1; else ibus[oo.inj_gen_p + i] = b;
    }
    for (int i = 0; i < g.n_load; ++i) { const int b = bus_of(e->h_load_sub[i], tp[e->h_load_pos[i]]); if (b >= 0) { act[b] = 1; ++n_other_at[b]; ibus[oo.inj_load_p + i] = b; } }
    for (int i = 0; i < g.n_sto; ++i) { const int b = bus_of(e->h_sto_sub[i], tp[e->h_sto_pos[i]]); if (b >= 0) { act[b] = 1; ++n_other_at[b]; ibus[oo.inj_sto_p + i] = b; } }
    for (int i = 0; i < nsh; ++i) { const int b = bus_of(e->h_shunt_sub[i], sbp[i]); if (b >= 0) { act[b] = 1; ++n_other_at[b]; ibus[oo.inj_sh_p + i] = b; } }
    for (int l = 0; l < nl; ++l) if (bf[l] >= 0) { ++n_lines_at[bf[l]]; ++n_lines_at[bt[l]]; }
    // compact numbering: active non-reference buses first, then the active reference buses
    compact.assign(nbt, -1);
    std::vector<int>& c2b = e->h_ptdfb_bus[c];
    c2b.reserve(nbt);
    int nr = 0, n_act = 0;
    for (int b = 0; b < nbt; ++b) if (act[b] && !ref[b]) { compact[b] = nr++; c2b.push_back(b); }
    n_act = nr;
    bool any_ref = false;
    for (int b = 0; b < nbt; ++b) if (act[b] && ref[b]) { compact[b] = n_act++; c2b.push_back(b); any_ref = true; }
    // connectivity (rundcpp(check_connectivity=True), pandaPowerBackend.py:1090): every active bus must reach a reference bus
    int status = any_ref ? 0 : 3;
    if (any_ref) {
      comp.resize(nbt);
      for (int b = 0; b < nbt; ++b) comp[b] = b;
      auto find = [&](int x) { while (comp[x] != x) { comp[x] = comp[comp[x]]; x = comp[x]; } return x; };
      for (int l = 0; l < nl; ++l) if (bf[l] >= 0 && bf[l] != bt[l]) comp[find(bf[l])] = find(bt[l]);
      has_ref.assign(nbt, 0);
      for (int b = 0; b < nbt; ++b) if (act[b] && ref[b]) has_ref[find(b)] = 1;
      for (int b = 0; b < nbt; ++b) if (act[b] && !has_ref[find(b)]) { status = 2; break; }
    }
    const int n_pad = std::max(16, (nr + 15) & ~15);
    if (n_pad > gpf::PTDFB_MAX_N) return 2;
    d[0] = nr; d[1] = n_act; d[2] = n_pad; d[3] = status;
    for (int l = 0; l < nl; ++l) {
      const bool on = bf[l] >= 0 && bf[l] != bt[l];
      lf[l] = on ? compact[bf[l]] : -1; lt[l] = on ? compact[bt[l]] : -1;
      // a line end on a bus that carries nothing else: the outage of the line removes the bus (no islanding, the other flows stand)
      lflag[l] = (on && ((n_lines_at[bf[l]] == 1 && n_other_at[bf[l]] == 0) || (n_lines_at[bt[l]] == 1 && n_other_at[bt[l]] == 0))) ? 1 : 0;
    }
    {   // rows of the reduced B': for every non-reference bus r the lines at it, ascending, as line | other end << 16
      int* cptr = lflag + nl;
      int* cent = cptr + gpf::PTDFB_MAX_N + 1;
      cnt.assign(nr + 1, 0);
      for (int l = 0; l < nl; ++l) { if (lf[l] < 0) continue; if (lf[l] < nr) ++cnt[lf[l]]; if (lt[l] < nr) ++cnt[lt[l]]; }
      int acc = 0;
      for (int r = 0; r < nr; ++r) { cptr[r] = acc; acc += cnt[r]; cnt[r] = cptr[r]; }
      for (int r = nr; r <= gpf::PTDFB_MAX_N; ++r) cptr[r] = acc;
      for (int i = 0; i < 2 * nl; ++i) cent[i] = 0;
      for (int l = 0; l < nl; ++l) {
        if (lf[l] < 0) continue;
        if (lf[l] < nr) cent[cnt[lf[l]]++] = l | (lt[l] << 16);
        if (lt[l] < nr) cent[cnt[lt[l]]++] = l | (lf[l] << 16);
      }
    }
    for (int i = 0; i < g.n_inj; ++i) ib[i] = ibus[i] >= 0 ? compact[ibus[i]] : -1;
    S_.npad_max = std::max(S_.npad_max, n_pad);
    S_.nact_max = std::max(S_.nact_max, n_act);
    return 0;
  };
  {
    // Descriptors are cached by topology row (hash + the row itself, compared on a hit): a rebuild after some lanes changed their topology
    // -- or the next contingency scan over the same family of topologies -- only walks the tables for classes it has not seen.
    ClsScratch scr;
    const size_t row_ints = (size_t)g.dim_topo + (size_t)nsh;
    const bool no_cache = getenv("GRIDPF_PTDFB_NO_CACHE") != nullptr;      // developer / bench: every class counts as never seen (read at every call)
    if (no_cache || e->ptdfb_cache_stride != stride || e->ptdfb_cache_n > 8192) { e->ptdfb_cache.clear(); e->ptdfb_cache_n = 0; e->ptdfb_cache_stride = stride; }
    std::vector<int> miss;
    for (int c = 0; c < nc; ++c) {
      const int* tp = topo.data() + (size_t)first_lane[c] * g.dim_topo;
      const int* sbp = sb.data() + (size_t)first_lane[c] * std::max(nsh, 1);
      int* d = desc.data() + (size_t)c * stride;
      auto it_b = e->ptdfb_cache.find(first_hash[c]);
      const gpf_engine::PtdfbCached* hit = nullptr;
      if (it_b != e->ptdfb_cache.end())
        for (const auto& ce : it_b->second)
          if (std::memcmp(ce.row.data(), tp, (size_t)g.dim_topo * sizeof(int)) == 0 && (!nsh || std::memcmp(ce.row.data() + g.dim_topo, sbp, (size_t)nsh * sizeof(int)) == 0)) { hit = &ce; break; }
      if (hit) {
        std::memcpy(d, hit->desc.data(), (size_t)stride * sizeof(int));
        e->h_ptdfb_bus[c] = hit->c2b;
      } else miss.push_back(c);
    }
    // the classes never seen before: their descriptors are independent table walks -- spread over the host threads
    std::vector<int> miss_err(miss.size(), 0);
    std::vector<gpf_engine::PtdfbCached> miss_ce(miss.size());      // (the cache entries too: three allocations + 9 KB of copies per class)
    (void)scr;
    par_for((int)miss.size(), 16, [&](int q0, int q1, int) {
      ClsScratch scr_t;
      for (int q = q0; q < q1; ++q) {
        const int c = miss[q];
        miss_err[q] = build_class(c, scr_t);
        if (miss_err[q]) continue;
        gpf_engine::PtdfbCached& ce = miss_ce[q];
        ce.row.resize(row_ints);
        std::memcpy(ce.row.data(), topo.data() + (size_t)first_lane[c] * g.dim_topo, (size_t)g.dim_topo * sizeof(int));
        if (nsh) std::memcpy(ce.row.data() + g.dim_topo, sb.data() + (size_t)first_lane[c] * nsh, (size_t)nsh * sizeof(int));
        const int* d = desc.data() + (size_t)c * stride;
        ce.desc.assign(d, d + stride);
        ce.c2b = e->h_ptdfb_bus[c];
      }
    });
    for (size_t q = 0; q < miss.size(); ++q) {
      const int c = miss[q];
      const int err = miss_err[q];
      if (err == 1) return fail(GPF_E_INVALID, "gpf_ptdf_build_batch: bus id out of range");
      if (err == 2) return fail(GPF_E_CAPACITY, "gpf_ptdf_build_batch: more than 256 active non-reference buses in one topology");
      e->ptdfb_cache[first_hash[c]].push_back(std::move(miss_ce[q]));
      ++e->ptdfb_cache_n;
    }
    for (int c = 0; c < nc; ++c) {
      const int* d = desc.data() + (size_t)c * stride;
      npad_max = std::max(npad_max, d[2]);
      nact_max = std::max(nact_max, d[1]);
    }
  }
  if (stage_timing) tm[3] = now_us();
  // ---- slots: lanes grouped by class, every group padded to a multiple of 16 ------------------------------------------------------------
  std::vector<std::vector<int>> members(nc);
  for (int k = 0; k < n; ++k) members[e->h_ptdfb_lane_class[k]].push_back(lane0 + k);
  for (int c = 0; c < nc; ++c) {
    for (int ln : members[c]) order.push_back(ln);
    while (order.size() & 15) order.push_back(-1);
    while (blk_class.size() * 16 < order.size()) blk_class.push_back(c);
  }
  n_slots = order.size();
  return GPF_OK;
  };
  if (dev) { nc = grp[0]; n_slots = (size_t)grp[1]; npad_max = grp[2]; nact_max = grp[3]; }
  else { const int rc_h = host_group(); if (rc_h != GPF_OK) return rc_h; }
  const int line_pad = (nl + 15) & ~15;
  const int nb_pad = std::max(4, (nact_max + 3) & ~3), kpad = (nb_pad + 31) & ~31;
  e->ptdf_ready = false;
  // (grow-only buffers: a rebuild after a few topology changes allocates nothing; uploads ride the engine's stream in front of the kernel)
  if (!dev) {
    HIP_TRY(e->ptdfb_desc.put(desc.data(), desc.size(), e->stream));
    HIP_TRY(e->ptdfb_order.put(order.data(), order.size(), e->stream));
    HIP_TRY(e->ptdfb_blk_class.put(blk_class.data(), blk_class.size(), e->stream));
  }
  HIP_TRY(e->ptdfb_status.ensure(nc));
  HIP_TRY(e->ptdfb_work.ensure((size_t)nc * npad_max * npad_max));
  HIP_TRY(e->ptdfb_t.ensure((size_t)nc * kpad * line_pad));
  if (with_lodf) HIP_TRY(e->ptdfb_lodf.ensure((size_t)nc * nl * line_pad)); else e->ptdfb_lodf.release();
  if (!e->ptdfb_inj_w.p) {
    std::vector<double> w(g.n_inj, 0.0);
    for (int i = 0; i < g.n_gen; ++i) w[oo.inj_gen_p + i] = 1.0;
    for (int i = 0; i < g.n_load; ++i) w[oo.inj_load_p + i] = -1.0;
    for (int i = 0; i < g.n_sto; ++i) w[oo.inj_sto_p + i] = -1.0;
    for (int i = 0; i < nsh; ++i) w[oo.inj_sh_p + i] = -e->h_shunt_fact[i];
    HIP_TRY(e->ptdfb_inj_w.upload(w.data(), w.size()));
  }
  if (e->ptdf_line_pad != line_pad || !e->ptdf_flow.p) { e->ptdf_flow.release(); HIP_TRY(e->ptdf_flow.alloc((size_t)e->cap_lanes * line_pad)); }
  if (with_lodf && (e->ptdf_line_pad != line_pad || !e->lodf_worst.p)) { e->lodf_worst.release(); HIP_TRY(e->lodf_worst.alloc((size_t)e->cap_lanes * line_pad)); }
  gpf::PtdfBuildDev D{};
  D.n_line = nl; D.line_pad = line_pad; D.n_inj = g.n_inj; D.kpad = kpad; D.desc_stride = stride;
  D.work_stride = (long long)npad_max * npad_max; D.ptdf_stride = (long long)kpad * line_pad; D.lodf_stride = (long long)nl * line_pad;
  D.desc = e->ptdfb_desc.p; D.br_bdc = e->br_bdc.p; D.work = e->ptdfb_work.p; D.ptdf_t = e->ptdfb_t.p; D.lodf = with_lodf ? e->ptdfb_lodf.p : nullptr;
  D.status = e->ptdfb_status.p;
  DevArr<long long> dbg;
  static const bool want_dbg = getenv("GRIDPF_PTDFB_DEBUG") != nullptr;     // developer: per-phase shader-clock stamps of class 0 on stderr
  if (want_dbg) { HIP_TRY(dbg.alloc((size_t)nc * 8)); HIP_TRY(hipMemset(dbg.p, 0, (size_t)nc * 8 * sizeof(long long))); D.dbg = dbg.p; }
  // reduced dimension <= 128 (118-substation grids): the matrix of a class lives in LDS (ptdf_build_lds_kernel), else in global memory
  const bool no_resident = getenv("GRIDPF_PTDFB_GLOBAL") != nullptr;   // developer / tests: force the global-memory kernel (read at every call)
  const bool resident = npad_max <= 128 && line_pad <= gpf::PTDFB_LDS_THREADS && !no_resident && gpf::ptdfb_lds_bytes_resident(npad_max, line_pad, nl) <= LDS_HARD_LIMIT;
  const size_t lds = resident ? gpf::ptdfb_lds_bytes_resident(npad_max, line_pad, nl) : gpf::ptdfb_lds_bytes(npad_max, line_pad);
  static size_t lds_set[64][2] = {{0}};
  if (lds > lds_set[e->device & 63][resident]) {
    HIP_TRY(hipFuncSetAttribute(resident ? reinterpret_cast<const void*>(&gpf::ptdf_build_lds_kernel) : reinterpret_cast<const void*>(&gpf::ptdf_build_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set[e->device & 63][resident] = lds;
  }
  if (stage_timing) tm[4] = now_us();
  if (!e->ptdfb_ev_a) { HIP_TRY(hipEventCreate(&e->ptdfb_ev_a)); HIP_TRY(hipEventCreate(&e->ptdfb_ev_b)); }     // (the engine's: destroyed with it)
  if (e->ptdfb_status_pin_n < (size_t)nc) {
    if (e->ptdfb_status_pin) (void)hipHostFree(e->ptdfb_status_pin);
    e->ptdfb_status_pin = nullptr; e->ptdfb_status_pin_n = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->ptdfb_status_pin), ((size_t)nc + nc / 4 + 64) * sizeof(int), hipHostMallocDefault));
    e->ptdfb_status_pin_n = (size_t)nc + nc / 4 + 64;
  }
  struct { hipEvent_t a, b; } ev{e->ptdfb_ev_a, e->ptdfb_ev_b};
  HIP_TRY(hipEventRecord(ev.a, e->stream));
  if (resident) hipLaunchKernelGGL(gpf::ptdf_build_lds_kernel, dim3(nc), dim3(gpf::PTDFB_LDS_THREADS), lds, e->stream, D);
  else hipLaunchKernelGGL(gpf::ptdf_build_kernel, dim3(nc), dim3(gpf::PTDFB_THREADS), lds, e->stream, D);
  hipError_t le = hipGetLastError();
  HIP_TRY(hipEventRecord(ev.b, e->stream));
  if (le != hipSuccess) return fail(GPF_E_DEVICE, std::string("ptdf_build_kernel: ") + hipGetErrorString(le));
  // the class status comes back by DMA into a pinned block behind the kernel; nobody waits here -- the flows / screening calls queue on the
  // same stream, gpf_ptdf_batch_info (status, kernel time) synchronises when it is asked (ptdfb_finish)
  e->h_ptdfb_status.assign(nc, 0);
  HIP_TRY(hipMemcpyAsync(e->ptdfb_status_pin, e->ptdfb_status.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  e->ptdfb_n = n; e->ptdfb_classes = nc; e->ptdfb_desc_stride = stride;      // (what ptdfb_finish / ptdfb_fetch_host size their copies by)
  e->ptdfb_host_stale = dev; e->ptdfb_bus_stale = dev;
  e->ptdfb_prefetched = false;
  if (dev) {                                        // what gpf_ptdf_batch_info will be asked for rides the same stream: one wait gets it all
    const size_t need = (size_t)n + (size_t)nc * 4;
    if (e->ptdfg_back_pin_n < need) {
      if (e->ptdfg_back_pin) (void)hipHostFree(e->ptdfg_back_pin);
      e->ptdfg_back_pin = nullptr; e->ptdfg_back_pin_n = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->ptdfg_back_pin), (need + need / 4 + 256) * sizeof(int), hipHostMallocDefault));
      e->ptdfg_back_pin_n = need + need / 4 + 256;
    }
    HIP_TRY(hipMemcpyAsync(e->ptdfg_back_pin, e->ptdfg_lane_class.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpy2DAsync(e->ptdfg_back_pin + n, 4 * sizeof(int), e->ptdfb_desc.p, (size_t)stride * sizeof(int), 4 * sizeof(int), (size_t)nc, hipMemcpyDeviceToHost, e->stream));
    e->ptdfb_prefetched = true;
  }
  e->ptdfb_pending = true;
  float ms = 0.f;
  if (stage_timing || want_dbg) { int rc_f = ptdfb_finish(e); if (rc_f != GPF_OK) return rc_f; ms = (float)e->ptdfb_kernel_ms; }
  if (stage_timing)
    fprintf(stderr, "[gridpf] ptdf_build_batch %d lanes, %d classes: rows to the host %.0f us, grouping %.0f, descriptors %.0f, slots + uploads %.0f, kernel + status %.0f\n",
            n, nc, tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], now_us() - tm[4]);
  if (want_dbg) {
    std::vector<long long> h((size_t)nc * 8);
    (void)hipMemcpy(h.data(), dbg.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    int c_ok = 0;
    while (c_ok < nc - 1 && e->h_ptdfb_status[c_ok] != 0) ++c_ok;
    const long long* s_ = h.data() + (size_t)c_ok * 8;
    fprintf(stderr, "[gridpf] ptdf_build_kernel class %d (n_pad %d), shader clocks: assemble %lld, gauss-jordan %lld (panel loads %lld, tile inversions %lld, trailing "
                    "updates %lld), PTDF^T %lld, LODF %lld (row builds %lld); kernel %.1f us\n", c_ok, dev ? 0 : desc[(size_t)c_ok * stride + 2], s_[1] - s_[0], s_[3] - s_[1], s_[7], s_[2], s_[6],
            s_[4] - s_[3], s_[5] ? s_[5] - s_[4] : 0LL, s_[7], ms * 1e3);
    dbg.release();
  }
  if (e->window) { ++e->win_launches; e->win_marked = false; }
  if (!dev) { e->h_ptdfb_desc = std::move(desc); e->h_ptdfb_hdr.clear(); }
  e->ptdfb_host_stale = dev; e->ptdfb_bus_stale = dev;                       // (device path: lane -> class map, descriptors, compact -> bus maps are fetched when somebody asks)
  e->ptdfb_lane0 = lane0; e->ptdfb_n = n; e->ptdfb_classes = nc; e->ptdfb_slots = (int)n_slots; e->ptdfb_kpad = kpad;
  e->ptdfb_npad_max = npad_max; e->ptdfb_desc_stride = stride;
  e->ptdf_nb_pad = nb_pad; e->ptdf_line_pad = line_pad;
  e->ptdf_rows_valid = 0;
  e->ptdf_batch = true;
  e->ptdf_ready = true;
  if (n_classes_out) *n_classes_out = nc;
  return GPF_OK;
}

int gpf_ptdf_batch_info(gpf_handle e, int32_t* lane_class, int32_t* class_status, int32_t* class_n, double* kernel_ms) {
  if (!e) return fail(GPF_E_INVALID, "gpf_ptdf_batch_info: null");
  if (!e->ptdf_ready || !e->ptdf_batch) return fail(GPF_E_INVALID, "gpf_ptdf_batch_info: call gpf_ptdf_build_batch first");
  if (class_status || kernel_ms) { const int rc_f = ptdfb_finish(e); if (rc_f != GPF_OK) return rc_f; }
  if (lane_class || class_n) { const int rc_m = ptdfb_fetch_host(e); if (rc_m != GPF_OK) return rc_m; }
  if (lane_class) std::copy(e->h_ptdfb_lane_class.begin(), e->h_ptdfb_lane_class.end(), lane_class);
  if (class_status) std::copy(e->h_ptdfb_status.begin(), e->h_ptdfb_status.end(), class_status);
  if (class_n) for (int c = 0; c < e->ptdfb_classes; ++c) class_n[c] = e->h_ptdfb_hdr.empty() ? e->h_ptdfb_desc[(size_t)c * e->ptdfb_desc_stride] : e->h_ptdfb_hdr[(size_t)c * 4];
  if (kernel_ms) *kernel_ms = e->ptdfb_kernel_ms;
  return GPF_OK;
}

int gpf_ptdf_batch_get(gpf_handle e, int32_t cls, double* ptdf, double* lodf) {
  if (!e) return fail(GPF_E_INVALID, "gpf_ptdf_batch_get: null");
  if (!e->ptdf_ready || !e->ptdf_batch) return fail(GPF_E_INVALID, "gpf_ptdf_batch_get: call gpf_ptdf_build_batch first");
  if (cls < 0 || cls >= e->ptdfb_classes) return fail(GPF_E_INVALID, "gpf_ptdf_batch_get: bad class");
  { const int rc_m = ptdfb_fetch_host(e, true); if (rc_m != GPF_OK) return rc_m; }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int nl = e->g.n_line, lp = e->ptdf_line_pad, nbt = e->g.nb_tot, kpad = e->ptdfb_kpad;
  if (ptdf) {
    std::vector<double> pt((size_t)kpad * lp);
    HIP_TRY(hipMemcpy(pt.data(), e->ptdfb_t.p + (size_t)cls * kpad * lp, pt.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::fill(ptdf, ptdf + (size_t)nl * nbt, 0.0);
    const std::vector<int>& c2b = e->h_ptdfb_bus[cls];
    for (size_t c = 0; c < c2b.size(); ++c)
      for (int l = 0; l < nl; ++l) ptdf[(size_t)l * nbt + c2b[c]] = pt[c * lp + l];
  }
  if (lodf) {
    if (!e->ptdfb_lodf.p) return fail(GPF_E_INVALID, "gpf_ptdf_batch_get: the batch was built without LODF tables");
    std::vector<float> lof((size_t)nl * lp);
    HIP_TRY(hipMemcpy(lof.data(), e->ptdfb_lodf.p + (size_t)cls * nl * lp, lof.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int m = 0; m < nl; ++m) for (int k = 0; k < nl; ++k) lodf[(size_t)m * nl + k] = (double)lof[(size_t)m * lp + k];
  }
  return GPF_OK;
}

int gpf_ptdf_get(gpf_handle e, double* ptdf) {
  if (!e || !ptdf) return fail(GPF_E_INVALID, "gpf_ptdf_get: null");
  if (!e->ptdf_ready || e->ptdf_batch) return fail(GPF_E_INVALID, "gpf_ptdf_get: call gpf_ptdf_build first (per-lane topologies: gpf_ptdf_batch_get)");
  std::copy(e->h_ptdf.begin(), e->h_ptdf.end(), ptdf);
  return GPF_OK;
}

int gpf_ptdf_flows(gpf_handle e, int32_t lane0, int32_t n) {
  if (!check_range(e, lane0, n)) return fail(GPF_E_INVALID, "gpf_ptdf_flows: bad range");
  if (!e->ptdf_ready) return fail(GPF_E_INVALID, "gpf_ptdf_flows: call gpf_ptdf_build first");
  if (n == 0) return GPF_OK;
  if (e->ptdf_batch && (lane0 != e->ptdfb_lane0 || n != e->ptdfb_n))
    return fail(GPF_E_INVALID, "gpf_ptdf_flows: per-lane topologies (gpf_ptdf_build_batch): the call must cover exactly the lane range that was built");
  HIP_TRY(hipSetDevice(e->device));
  const gpf::PtdfDev P = ptdf_dev(e);
  const size_t lds_a = (size_t)16 * gpf::ptdf_a_stride(P.nb_pad) * sizeof(double);
  const int n_blk = e->ptdf_batch ? e->ptdfb_slots / 16 : (n + 15) / 16;
  hipLaunchKernelGGL(gpf::ptdf_flows_kernel, dim3(n_blk, (P.line_pad / 16 + 3) / 4), dim3(256), lds_a, e->stream, P, e->inj.p, lane0, n,
                     e->ptdf_flow.p);
  HIP_TRY(hipGetLastError());
  if (e->window) { ++e->win_launches; e->win_marked = false; }
  return GPF_OK;
}

int gpf_ptdf_flows_rows(gpf_handle e, int32_t t0, int32_t n_rows, double rebalance) {
  if (!e || n_rows <= 0) return fail(GPF_E_INVALID, "gpf_ptdf_flows_rows: bad arguments");
  if (!e->ptdf_ready) return fail(GPF_E_INVALID, "gpf_ptdf_flows_rows: call gpf_ptdf_build first");
  if (!e->chron.p || e->chron_T <= 0) return fail(GPF_E_INVALID, "gpf_ptdf_flows_rows: no chronics uploaded");
  HIP_TRY(hipSetDevice(e->device));
  const size_t need = (size_t)n_rows * e->cap_lanes * e->ptdf_line_pad;
  if (e->ptdf_flow_rows.n < need) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->ptdf_flow_rows.release();
    HIP_TRY(e->ptdf_flow_rows.alloc(need));
  }
  if (e->ptdf_batch && (e->ptdfb_lane0 != 0 || e->ptdfb_n != e->n_lanes))
    return fail(GPF_E_INVALID, "gpf_ptdf_flows_rows: per-lane topologies (gpf_ptdf_build_batch) must have been built for ALL lanes");
  const gpf::PtdfDev P = ptdf_dev(e);
  gpf::PtdfRowsDev R{};
  R.chron = e->chron.p; R.lane_table = e->lane_table.p; R.lane_offset = e->lane_offset.p;
  R.lane_scale = e->has_scale ? e->lane_scale.p : nullptr; R.lane_gen_delta = e->has_delta ? e->lane_gen_delta.p : nullptr;
  R.gen_slack = e->gen_slack.p; R.T = e->chron_T; R.n_chron = e->g.n_chron; R.n_load = e->g.n_load; R.n_gen = e->g.n_gen;
  R.inj_gen_p = e->oo.inj_gen_p; R.inj_load_p = e->oo.inj_load_p; R.inj_sto_p = e->oo.inj_sto_p; R.n_inj_tail = e->g.n_inj - e->oo.inj_sto_p;
  R.rebalance = rebalance;
  R.kpad = (P.nb_pad + 31) & ~31;
  static const int mt_env = std::getenv("GRIDPF_PTDF_MT") ? std::atoi(std::getenv("GRIDPF_PTDF_MT")) : 0;      // developer override (1 | 2 | 4)
  const int mt = (mt_env == 1 || mt_env == 2 || mt_env == 4) ? mt_env : 2;
  const int NP = 16 * mt;
  const size_t lds_a = (size_t)NP * gpf::ptdf_rows_stride(R.kpad) * sizeof(double);
  const long long n_pairs = (long long)e->n_lanes * n_rows;
  // per-lane topologies: one block per (group of 16 slots, mt consecutive rows), see ptdf_rows_kernel
  const int n_units = e->ptdf_batch ? e->ptdfb_slots : e->n_lanes;
  const dim3 grid(e->ptdf_batch ? (unsigned)((size_t)(e->ptdfb_slots / 16) * ((n_rows + mt - 1) / mt)) : (unsigned)((n_pairs + NP - 1) / NP));
  static size_t lds_set[64][3] = {{0}};
#define GPF_PTDF_ROWS(MT_, SLOT_)                                                                                                      \
  do {                                                                                                                                 \
    if (lds_a > lds_set[e->device & 63][SLOT_]) {                                                                                      \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gpf::ptdf_rows_kernel<MT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)); \
      lds_set[e->device & 63][SLOT_] = lds_a;                                                                                          \
    }                                                                                                                                  \
    hipLaunchKernelGGL(gpf::ptdf_rows_kernel<MT_>, grid, dim3(256), lds_a, e->stream, P, R, e->inj.p, n_units, (long long)e->cap_lanes, t0, \
                       n_rows, e->ptdf_flow_rows.p);                                                                                   \
  } while (0)
  if (mt == 4) GPF_PTDF_ROWS(4, 2); else if (mt == 2) GPF_PTDF_ROWS(2, 1); else GPF_PTDF_ROWS(1, 0);
#undef GPF_PTDF_ROWS
  HIP_TRY(hipGetLastError());
  e->ptdf_rows_valid = n_rows;
  if (e->window) { ++e->win_launches; e->win_marked = false; }
  return GPF_OK;
}

int gpf_get_ptdf_flows_rows(gpf_handle e, int32_t row0, int32_t n_rows, int32_t lane0, int32_t n, float* p_or) {
  if (!check_range(e, lane0, n) || !p_or || row0 < 0 || n_rows < 0 || row0 + n_rows > e->ptdf_rows_valid)
    return fail(GPF_E_INVALID, "gpf_get_ptdf_flows_rows: bad range (only the rows of the last gpf_ptdf_flows_rows are retrievable)");
  HIP_TRY(hipSetDevice(e->device));
  for (int r = 0; r < n_rows; ++r)
    HIP_TRY(hipMemcpy2DAsync(p_or + (size_t)r * n * e->g.n_line, (size_t)e->g.n_line * sizeof(float),
                             e->ptdf_flow_rows.p + ((size_t)(row0 + r) * e->cap_lanes + lane0) * e->ptdf_line_pad,
                             (size_t)e->ptdf_line_pad * sizeof(float), (size_t)e->g.n_line * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_get_ptdf_flows(gpf_handle e, int32_t lane0, int32_t n, float* p_or) {
  if (!check_range(e, lane0, n) || !p_or) return fail(GPF_E_INVALID, "gpf_get_ptdf_flows: bad arguments");
  if (!e->ptdf_ready) return fail(GPF_E_INVALID, "gpf_get_ptdf_flows: call gpf_ptdf_build first");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpy2DAsync(p_or, (size_t)e->g.n_line * sizeof(float), e->ptdf_flow.p + (size_t)lane0 * e->ptdf_line_pad,
                           (size_t)e->ptdf_line_pad * sizeof(float), (size_t)e->g.n_line * sizeof(float), (size_t)n,
                           hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

int gpf_lodf_screen(gpf_handle e, int32_t lane0, int32_t n, const float* cap_mw, float* worst) {
  if (!check_range(e, lane0, n) || !worst) return fail(GPF_E_INVALID, "gpf_lodf_screen: bad arguments");
  if (!e->ptdf_ready) return fail(GPF_E_INVALID, "gpf_lodf_screen: call gpf_ptdf_build and gpf_ptdf_flows first");
  if (n == 0) return GPF_OK;
  if (e->ptdf_batch && (lane0 != e->ptdfb_lane0 || n != e->ptdfb_n || !e->ptdfb_lodf.p))
    return fail(GPF_E_INVALID, "gpf_lodf_screen: per-lane topologies: build them with LODF tables and screen exactly the lane range that was built");
  HIP_TRY(hipSetDevice(e->device));
  const int nl = e->g.n_line, lp = e->ptdf_line_pad;
  const float* ic = nullptr;
  if (cap_mw) {
    std::vector<float> inv(nl);
    for (int l = 0; l < nl; ++l) inv[l] = cap_mw[l] > 0.f ? 1.0f / cap_mw[l] : 0.f;
    if (!e->lodf_inv_cap.p) HIP_TRY(e->lodf_inv_cap.alloc(nl));
    HIP_TRY(hipMemcpyAsync(e->lodf_inv_cap.p, inv.data(), (size_t)nl * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    ic = e->lodf_inv_cap.p;
  }
  const size_t lds = ((size_t)5 * gpf::LODF_LPW + 1) * lp * sizeof(float);
  if (e->ptdf_batch)
    hipLaunchKernelGGL(gpf::lodf_screen_kernel, dim3(e->ptdfb_slots / gpf::LODF_LPW), dim3(256), lds, e->stream, nl, lp, e->ptdfb_lodf.p, ic, e->ptdf_flow.p,
                       lane0, n, e->lodf_worst.p, e->ptdfb_order.p, e->ptdfb_blk_class.p, (long long)nl * lp, e->ptdfb_status.p);
  else
    hipLaunchKernelGGL(gpf::lodf_screen_kernel, dim3((n + gpf::LODF_LPW - 1) / gpf::LODF_LPW), dim3(256), lds, e->stream, nl, lp, e->lodf.p, ic,
                       e->ptdf_flow.p, lane0, n, e->lodf_worst.p, nullptr, nullptr, 0LL, nullptr);
  HIP_TRY(hipGetLastError());
  if (e->window) { ++e->win_launches; e->win_marked = false; }
  HIP_TRY(hipMemcpy2DAsync(worst, (size_t)nl * sizeof(float), e->lodf_worst.p + (size_t)lane0 * lp, (size_t)lp * sizeof(float),
                           (size_t)nl * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return GPF_OK;
}

#ifdef GPF_TIMING
int gpf_debug_read_work(gpf_handle e, double* out, int64_t n) {
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(out, e->work.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return GPF_OK;
}
#endif

int gpf_get_counters(gpf_handle e, int64_t out[2]) {
  if (!e || !out) return fail(GPF_E_INVALID, "gpf_get_counters: null");
  out[0] = e->n_step_calls; out[1] = e->n_step_dispatches;
  return GPF_OK;
}

int gpf_get_plan(gpf_handle e, int32_t out[8]) {
  if (!e || !out) return fail(GPF_E_INVALID, "gpf_get_plan: null");
  LaunchPlan p, pb;
  int rc = plan_launch(e, 0, e->n_lanes, p, pb);
  if (rc != GPF_OK) return rc;
  out[0] = p.sparse_nb; out[1] = p.ipw; out[2] = p.wpi; out[3] = p.sparse_stage; out[4] = p.yreg ? 1 : 0; out[5] = p.dcf;
  out[6] = (int32_t)p.lds; out[7] = p.tc ? 1 : 0;
  return GPF_OK;
}

int gpf_device_pointers(gpf_handle e, void** ptrs, void** stream) { return gpf_device_pointers_n(e, ptrs, GPF_N_DEVICE_POINTERS, stream); }

int gpf_device_pointers_n(gpf_handle e, void** out, int32_t n_ptrs, void** stream) {
  if (!e || !out || n_ptrs < 0) return fail(GPF_E_INVALID, "gpf_device_pointers: null");
  void* ptrs[GPF_N_DEVICE_POINTERS];
  ptrs[0] = e->inj.p; ptrs[1] = e->topo.p; ptrs[2] = e->shunt_bus.p; ptrs[3] = e->out.p; ptrs[4] = e->topo_out.p;
  ptrs[5] = e->line_status.p; ptrs[6] = e->status.p; ptrs[7] = e->chron.p;
  ptrs[8] = e->rho.p; ptrs[9] = e->overflow_count.p; ptrs[10] = e->done.p; ptrs[11] = e->episode.p; ptrs[12] = e->bus_vm.p;
  ptrs[13] = e->bus_va.p; ptrs[14] = e->shunt_bus_out.p; ptrs[15] = e->disc_round.p;
  ptrs[16] = e->traj_cap ? e->traj_rho.p : nullptr; ptrs[17] = e->traj_cap ? e->traj_status.p : nullptr;
  const bool obs = e->traj_cap && (e->traj_what & GPF_TRAJ_OBS);
  ptrs[18] = obs ? e->traj_out.p : nullptr; ptrs[19] = obs ? e->traj_topo.p : nullptr; ptrs[20] = obs ? e->traj_shb.p : nullptr;
  ptrs[21] = obs ? e->traj_lstat.p : nullptr;
  ptrs[22] = e->env_on ? e->env_act_redisp.p : nullptr; ptrs[23] = e->env_on ? e->env_act_storage.p : nullptr;
  ptrs[24] = e->env_on ? e->env_act_curtail.p : nullptr; ptrs[25] = e->env_on ? e->env_target.p : nullptr;
  ptrs[26] = e->env_on ? e->env_actual.p : nullptr; ptrs[27] = e->env_on ? e->env_charge.p : nullptr;
  for (int i = 0; i < n_ptrs; ++i) out[i] = i < GPF_N_DEVICE_POINTERS ? ptrs[i] : nullptr;     // never writes beyond the caller's array
  if (stream) *stream = e->stream;
  return GPF_OK;
}

}  // extern "C"
