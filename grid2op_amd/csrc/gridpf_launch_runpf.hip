// gridpf_launch_runpf.hip -- template instantiations + dispatch of gpf::runpf_sparse_kernel (kernel S, one power flow per lane).
#include "gridpf_host.hpp"

namespace {

template <int NB, int ST, int IPW, int WP, bool TC, bool YR = false>
hipError_t launch(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int lane0, int n_l,
                  const int* list, int is_dc, int max_iter, double tol_pu) {
  static size_t lds_set[64] = {0};
  if (p.jit && p.jit->on) {                  // grid-specialised kernel of this variant (gridpf_jit.hip), compiled on first use
    if (hipFunction_t f = gpf_jit_get(*p.jit, NB, ST, IPW, WP, TC, YR, false, true, p.lds)) {
      const int* cls = p.cls_list;
      void* args[] = {(void*)&d_params, (void*)&lane0, (void*)&list, (void*)&cls, (void*)&is_dc, (void*)&max_iter, (void*)&tol_pu};
      ++p.jit->n_launches;
      return hipModuleLaunchKernel(f, (unsigned)((n_l + IPW - 1) / IPW), 1, 1, (unsigned)(gpf::WAVE * WP), 1, 1, (unsigned)p.lds, stream, args, nullptr);
    }
  }
  auto kern = &gpf::runpf_sparse_kernel<NB, ST, IPW, 2, WP, TC, YR>;
  if (p.lds > lds_set[device & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    if (e != hipSuccess) return e;
    lds_set[device & 63] = p.lds;
  }
  hipLaunchKernelGGL(kern, dim3((n_l + IPW - 1) / IPW), dim3(gpf::WAVE * WP), p.lds, stream, d_params, lane0, list, p.cls_list, is_dc,
                     max_iter, tol_pu);
  return hipGetLastError();
}

}  // namespace

// one launch for a contiguous range, or for the lanes of a device index list
hipError_t gpf_launch_runpf_sparse(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int lane0, int n,
                                   int is_dc, int max_iter, double tol_pu) {
  const int n_l = p.n_list ? p.n_list : n;
  const int* list = p.n_list ? p.list : nullptr;
#define LAUNCH_ARGS p, device, d_params, stream, lane0, n_l, list, is_dc, max_iter, tol_pu
#define GO(NB, ST, IPW, WP, TC) return launch<NB, ST, IPW, WP, TC>(LAUNCH_ARGS)
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
#undef LAUNCH_ARGS
  return hipErrorInvalidValue;
}
