// gridpf_host.hpp -- host-side declarations shared by the translation units of libgridpf.so: the launch plan of kernel S and
// the two dispatchers (the template instantiations of the power-flow kernel and of the step kernel are compiled in
// separate translation units so that they build in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "gridpf_sparse.hpp"

// Run-time specialisation of the step kernel for the engine's grid (gridpf_jit.hip); one per engine, off by default.
struct GpfJit {
  bool on = false;
  std::string header;                       // gpf_jit_header() of the engine's parameter block at gpf_jit_enable
  std::string src_dir, cache_dir, hipcc, flags_user;   // flags_user: GRIDPF_JIT_FLAGS (experiments), empty: default policy
  std::string aot_dir, arch, defines, compile_why;     // ahead-of-time code objects ("_aot" next to the library; empty: none), device arch, -D flags of the library build
  uint64_t src_hash = 0, aot_hash = 0;      // cache key seeds: sources + compiler version / sources only (ahead-of-time objects)
  bool can_compile = false;                 // a private cache directory and a compiler that runs
  int n_aot = 0;                            // variants loaded from the ahead-of-time directory
  std::map<unsigned, hipFunction_t> fns;    // kernel variant -> specialised kernel (nullptr: failed, ahead-of-time kernel used)
  std::vector<hipModule_t> mods;
  std::string variants, message;            // "<1,2,2,2,1,false,false,false> ..." loaded so far; last error
  int n_compiled = 0, n_cached = 0, n_failed = 0;
  long long n_launches = 0;                 // launches that went through a specialised kernel
  double seconds = 0.0;                     // time spent compiling / loading
};
std::string gpf_jit_header(const gpf::DevParamsS& hp);
int gpf_jit_configure(GpfJit& j, const char* src_dir, const char* cache_dir, std::string& err);
bool gpf_jit_has_aot(const GpfJit& j, const std::string& header);
void gpf_jit_release(GpfJit& j);
hipFunction_t gpf_jit_get(GpfJit& j, int NB, int ST, int IPW, int WP, bool TC, bool YR, bool ENV, bool runpf, size_t lds_bytes);

struct LaunchPlan {
  size_t lds;
  int sparse_nb;    // 0: no launch; 1..3: block-sparse kernel S with NB busbars per substation block
  int minw;         // kernel S: __launch_bounds__ waves per SIMD (4 caps the kernel at 128 VGPRs: only worth it when LDS allows > 8 blocks per CU)
  bool tc;          // topology-class launch: single-busbar kernel on the bus-level graph of each lane's class (cls_list)
  const int* cls_list;
  int tc_rows, tc_nslot, tc_nslot_y;
  int n_list;       // > 0: this plan covers n_list lanes given by a device index list (mixed batches), else a contiguous range
  const int* list;  // device pointer (padded with a ghost lane to a multiple of ipw)
  int wpi;          // kernel S: wavefronts per instance (1, 2 or 4; > 1 only for NB == 1, IPW == 1 on large grids)
  int ipw;          // kernel S: grid instances per wavefront (1, 2 or 4; > 1 only for NB == 1 on small grids)
  bool env;          // step launches: the ENV instantiation (environment injection dynamics on): tables in global memory unless instance groups
  bool yreg;         // Ybus blocks in registers (gridpf_sparse.hpp: YR): NB == 1, 2 wavefronts per instance, tables in global memory
  int dcf;           // the LDS layout of this launch has room for the factored DC matrix (DevParamsS::dcf)
  GpfJit* jit;       // grid-specialised kernels of the engine (nullptr / !on: ahead-of-time kernels)
  int sparse_stage;  // 0: static tables read in place (L2), 1: program + pair table + injection row in LDS, 2: everything in LDS // program staged in LDS (small grids) or streamed from L2 (keeps 3 instances per CU on 118-bus grids)
};


// dispatch of the kernel-S variants (gridpf_launch_runpf.hip / gridpf_launch_step.hip); they return the HIP status of the launch
hipError_t gpf_launch_runpf_sparse(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int lane0, int n,
                                   int is_dc, int max_iter, double tol_pu);
// (*n_dispatched += the kernel dispatches issued: a batch of a few residency rounds goes out as one dispatch per round)
hipError_t gpf_launch_step_sparse(const LaunchPlan& p, int device, const gpf::DevParamsS* d_params, hipStream_t stream, int n_lanes,
                                  int max_iter, double tol_pu, const gpf::StepArgs& sa, int* n_dispatched = nullptr);
