// gridpf_launch_step.hip -- template instantiations + dispatch of gpf::step_sparse_kernel (kernel S, batched env steps).
#include "gridpf_host.hpp"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

// Resident blocks of a kernel on the whole device (blocks per CU x CUs), cached per kernel and LDS size; 0: unknown.
struct Residency { const void* key; size_t lds; int device, slots; };
int resident_slots(const void* kern, hipFunction_t fn, int device, int threads, size_t lds) {
  static std::vector<Residency> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  const void* key = fn ? reinterpret_cast<const void*>(fn) : kern;
  for (const Residency& r : cache) if (r.key == key && r.lds == lds && r.device == device) return r.slots;
  int per_cu = 0, n_cu = 0;
  hipError_t e = fn ? hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds);
  if (e != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
  cache.push_back({key, lds, device, per_cu * n_cu});
  return per_cu * n_cu;
}

// Dispatches of one step launch over n_vb instance groups.  Normally ONE: the hardware dispatcher refills the CUs as blocks retire.  A
// batch of a FEW residency rounds of equal lanes (2 048 lanes of a 118-substation grid = 2 rounds of the 1 024 resident blocks) is cut
// into that many back-to-back dispatches of equal size on the same stream: the dispatcher's greedy refill let the faster slots of a CU take
// a third block that then ran beyond the slower slots' second one -- 2.57 ms instead of 2.06 ms in half of the launches, +0.5 block
// durations at 3 and 4 rounds too (tools/exp_variance.py, round 5) -- while whole rounds in sequence cost 2 x 1.006 ms, every time.
// GRIDPF_SPLIT_ROUNDS = the largest number of rounds handled this way (default 4; 0: never).  Batches of many rounds keep the single
// dispatch: their lanes may differ in cost (N-1 contingencies that diverge) and only the dispatcher balances that.
int dispatches_for(int n_vb, int slots, bool tc) {
  static const int max_rounds = getenv("GRIDPF_SPLIT_ROUNDS") ? atoi(getenv("GRIDPF_SPLIT_ROUNDS")) : 4;
  if (tc || max_rounds <= 0 || slots <= 0 || n_vb <= slots || (long long)n_vb > (long long)max_rounds * slots) return 1;
  return (n_vb + slots - 1) / slots;
}

template <int NB, int ST, int IPW, int WP, bool TC, bool YR = false, bool ENV = false>
hipError_t launch(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int n_l, const int* list,
                  int max_iter, double tol_pu, const gpf::StepArgs& sa_in, int* n_dispatched) {
  static size_t lds_set[64] = {0};
  static const size_t pad = getenv("GRIDPF_LDS_PAD") ? (size_t)atoi(getenv("GRIDPF_LDS_PAD")) : 0;   // occupancy experiments only
  auto kern = &gpf::step_sparse_kernel<NB, ST, IPW, 2, WP, TC, YR, ENV>;
  const size_t lds = p.lds + pad;
  const int n_vb = (n_l + IPW - 1) / IPW;
  hipFunction_t f = nullptr;
  if (p.jit && p.jit->on) f = gpf_jit_get(*p.jit, NB, ST, IPW, WP, TC, YR, ENV, false, lds);   // grid-specialised kernel (gridpf_jit.hip), compiled on first use
  if (!f && lds > lds_set[device & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set[device & 63] = lds;
  }
  const int n_disp = dispatches_for(n_vb, resident_slots(reinterpret_cast<const void*>(kern), f, device, gpf::WAVE * WP, lds), TC);
  const int per = (n_vb + n_disp - 1) / n_disp;
  if (f) ++p.jit->n_launches;
  for (int d = 0, vb0 = 0; d < n_disp && vb0 < n_vb; ++d, vb0 += per) {
    const int nb = std::min(per, n_vb - vb0);
    gpf::StepArgs sa = sa_in;
    sa.lane0 = sa_in.lane0 + vb0 * IPW;                       // contiguous launches: first lane of this dispatch
    const int* lst = list ? list + (size_t)vb0 * IPW : nullptr;   // list launches: its part of the (ghost-padded) lane list
    if (f) {
      const int* cls = p.cls_list;
      void* args[] = {(void*)&d_params, (void*)&lst, (void*)&cls, (void*)&max_iter, (void*)&tol_pu, (void*)&sa};
      hipError_t e = hipModuleLaunchKernel(f, (unsigned)nb, 1, 1, (unsigned)(gpf::WAVE * WP), 1, 1, (unsigned)lds, stream, args, nullptr);
      if (e != hipSuccess) return e;
    } else {
      hipLaunchKernelGGL(kern, dim3(nb), dim3(gpf::WAVE * WP), lds, stream, d_params, lst, p.cls_list, max_iter, tol_pu, sa);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    if (n_dispatched) ++*n_dispatched;
  }
  return hipSuccess;
}

}  // namespace

// one launch for all the lanes of the engine, or for the lanes of a device index list
hipError_t gpf_launch_step_sparse(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int n_lanes,
                                  int max_iter, double tol_pu, const gpf::StepArgs& sa, int* n_dispatched) {
  const int n_l = p.n_list ? p.n_list : n_lanes;
  const int* list = p.n_list ? p.list : nullptr;
#define LAUNCH_ARGS p, device, d_params, stream, n_l, list, max_iter, tol_pu, sa, n_dispatched
#define GO(NB, ST, IPW, WP, TC) return launch<NB, ST, IPW, WP, TC>(LAUNCH_ARGS)
#define GOE(ST, IPW, WP, TC) return launch<1, ST, IPW, WP, TC, false, true>(LAUNCH_ARGS)
  if (p.env) {          // environment injection dynamics: single-busbar kernels, tables in LDS only with instance groups (plan_launch)
    if (p.sparse_nb != 1) return hipErrorInvalidValue;
    if (p.tc) {
      if (p.ipw == 4) GOE(0, 4, 1, true);
      if (p.ipw == 2) GOE(0, 2, 1, true);
      if (p.wpi == 2) GOE(0, 1, 2, true);
      GOE(0, 1, 1, true);
    }
    if (p.ipw == 4) GOE(2, 4, 1, false);
    if (p.ipw == 2) GOE(2, 2, 1, false);
    if (p.wpi == 2) {
      if (p.yreg) return launch<1, 0, 1, 2, false, true, true>(LAUNCH_ARGS);
      GOE(0, 1, 2, false);
    }
    GOE(0, 1, 1, false);
  }
  if (p.tc) {
    if (p.ipw == 4) GO(1, 0, 4, 1, true);
    if (p.ipw == 2) GO(1, 0, 2, 1, true);
    if (p.wpi == 2) GO(1, 0, 1, 2, true);
    GO(1, 0, 1, 1, true);
  }
  if (p.sparse_nb == 1) {
    if (p.ipw == 4) GO(1, 2, 4, 1, false);
    if (p.ipw == 2) GO(1, 2, 2, 1, false);
    if (p.wpi == 2) {
      if (p.sparse_stage == 2) GO(1, 2, 1, 2, false);
      if (p.sparse_stage == 1) GO(1, 1, 1, 2, false);
      if (p.yreg) return launch<1, 0, 1, 2, false, true>(LAUNCH_ARGS);
      GO(1, 0, 1, 2, false);
    }
    if (p.sparse_stage == 2) GO(1, 2, 1, 1, false);
    if (p.sparse_stage == 1) GO(1, 1, 1, 1, false);
    GO(1, 0, 1, 1, false);
  }
  if (p.sparse_nb == 2) {
    if (p.wpi == 2) { if (p.sparse_stage) GO(2, 1, 1, 2, false); GO(2, 0, 1, 2, false); }
    if (p.sparse_stage) GO(2, 1, 1, 1, false);
    GO(2, 0, 1, 1, false);
  }
  if (p.sparse_nb == 3) {
    if (p.sparse_stage) GO(3, 1, 1, 1, false);
    GO(3, 0, 1, 1, false);
  }
#undef GO
#undef GOE
#undef LAUNCH_ARGS
  return hipErrorInvalidValue;
}
