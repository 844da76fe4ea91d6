#!/bin/bash
# Developer tool: build libgridpf from the WORKING TREE with extra hipcc flags into grid2op_amd/libgridpf_<name>.so (same-box A/B
# of compile-time experiments; GRIDPF_LIB selects the library).  usage: tools/build_worktree_variant.sh <name> [extra hipcc flags]
name=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
W=/tmp/gpf_wt_$name
rm -rf $W; mkdir -p $W
for u in gridpf_capi gridpf_launch_runpf gridpf_launch_step; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $R/grid2op_amd/csrc/$u.hip -o $W/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $W/*.o -o $R/grid2op_amd/libgridpf_$name.so && echo built $R/grid2op_amd/libgridpf_$name.so
