// gridpf_launch_step.hip -- template instantiations + dispatch of gpf::step_sparse_kernel (kernel S, batched env steps).
#include "gridpf_host.hpp"
#include <cstdlib>

namespace {

template <int NB, int ST, int IPW, int WP, bool TC, bool YR = false, bool ENV = false>
hipError_t launch(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int n_l, const int* list,
                  int max_iter, double tol_pu, const gpf::StepArgs& sa) {
  static size_t lds_set[64] = {0};
  static const size_t pad = getenv("GRIDPF_LDS_PAD") ? (size_t)atoi(getenv("GRIDPF_LDS_PAD")) : 0;   // occupancy experiments only
  auto kern = &gpf::step_sparse_kernel<NB, ST, IPW, 2, WP, TC, YR, ENV>;
  const size_t lds = p.lds + pad;
  if (p.jit && p.jit->on) {                  // grid-specialised kernel of this variant (gridpf_jit.hip), compiled on first use
    if (hipFunction_t f = gpf_jit_get(*p.jit, NB, ST, IPW, WP, TC, YR, ENV, false, lds)) {
      const int* cls = p.cls_list;
      void* args[] = {(void*)&d_params, (void*)&list, (void*)&cls, (void*)&max_iter, (void*)&tol_pu, (void*)&sa};
      ++p.jit->n_launches;
      return hipModuleLaunchKernel(f, (unsigned)((n_l + IPW - 1) / IPW), 1, 1, (unsigned)(gpf::WAVE * WP), 1, 1, (unsigned)lds, stream, args, nullptr);
    }
  }
  if (lds > lds_set[device & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set[device & 63] = lds;
  }
  hipLaunchKernelGGL(kern, dim3((n_l + IPW - 1) / IPW), dim3(gpf::WAVE * WP), lds, stream, d_params, list, p.cls_list, max_iter, tol_pu,
                     sa);
  return hipGetLastError();
}

}  // namespace

// one launch for all the lanes of the engine, or for the lanes of a device index list
hipError_t gpf_launch_step_sparse(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int n_lanes,
                                  int max_iter, double tol_pu, const gpf::StepArgs& sa) {
  const int n_l = p.n_list ? p.n_list : n_lanes;
  const int* list = p.n_list ? p.list : nullptr;
#define LAUNCH_ARGS p, device, d_params, stream, n_l, list, max_iter, tol_pu, sa
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
