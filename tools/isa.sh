#!/bin/bash
# Developer tool (no GPU needed): ISA of ONE grid-specialised kernel variant, compiled like __graft_entry__.build_aot does.
#   tools/isa.sh <aot header> <step|runpf> '<variant>' <out prefix> [extra hipcc flags]
R=$(cd $(dirname $0)/.. && pwd); hdr=$1; kind=$2; var=$3; out=$4; shift 4
src=$out.hip
{ echo '#include <hip/hip_runtime.h>'; echo '#include "gridpf_common.hpp"'; echo '#include "gridpf_sparse.hpp"'; echo 'namespace gpf {'
  if [ $kind = runpf ]; then echo "template __global__ void runpf_sparse_kernel<$var>(const DevParamsS* __restrict__, int, const int* __restrict__, const int* __restrict__, int, int, double);"
  else echo "template __global__ void step_sparse_kernel<$var>(const DevParamsS* __restrict__, const int* __restrict__, const int* __restrict__, int, double, StepArgs);"; fi
  echo '}'; } > $src
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$@" -DGPF_JIT -include $hdr -I$R/grid2op_amd/csrc $src -o $out.s || exit 1
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size)" $out.s
wc -l $out.s
