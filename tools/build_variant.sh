#!/bin/bash
# Developer tool: build libgridpf from another git revision (default HEAD) into grid2op_amd/libgridpf_<name>.so for same-box A/B
# runs (GRIDPF_LIB selects the library).  usage: tools/build_variant.sh <name> [rev] [extra hipcc flags]
name=$1; rev=${2:-HEAD}; shift; shift
R=$(cd $(dirname $0)/.. && pwd)
W=/tmp/gpf_variant_$name
rm -rf $W; mkdir -p $W
git -C $R archive $rev grid2op_amd/csrc include | tar -x -C $W
for u in gridpf_capi gridpf_launch_runpf gridpf_launch_step; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $W/grid2op_amd/csrc/$u.hip -o $W/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $W/*.o -o $R/grid2op_amd/libgridpf_$name.so && echo built $R/grid2op_amd/libgridpf_$name.so
