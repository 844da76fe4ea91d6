#!/bin/bash
# Developer tool: a self-contained build of another git revision (default HEAD) for same-box A/B runs: tools/_build/<name>/grid2op_amd/libgridpf.so
# with ITS kernel sources beside it (.../grid2op_amd/csrc: the run-time specialisation compiles what lies next to the library).
#   tools/build_base.sh <name> [rev];   GRIDPF_LIB=$PWD/tools/_build/<name>/grid2op_amd/libgridpf.so python bench.py ...
name=$1; rev=${2:-HEAD}
R=$(cd $(dirname $0)/.. && pwd); W=$R/tools/_build/$name
rm -rf $W; mkdir -p $W/obj
git -C $R archive $rev grid2op_amd/csrc include | tar -x -C $W
for u in gridpf_capi gridpf_jit gridpf_launch_runpf gridpf_launch_step; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $W/grid2op_amd/csrc/$u.hip -o $W/obj/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $W/obj/*.o -o $W/grid2op_amd/libgridpf.so && rm -rf $W/obj && echo built $W/grid2op_amd/libgridpf.so
