#!/bin/bash
# VGPRs / scratch / spills of every kernel instantiation of one translation unit (default: the step kernels)
unit=${1:-grid2op_amd/csrc/gridpf_launch_step.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA -Rpass-analysis=kernel-resource-usage -c "$unit" -o /tmp/_kr.o 2>&1 |
  grep "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/^Function Name/ {if (n) print n, v, sg, sc, sp; n=$3} /^VGPRs:/ {v="vgpr="$2} /^TotalSGPRs:/ {sg="sgpr="$2} /^ScratchSize/ {sc="scratch="$3} /^SGPRs Spill/ {sp="sgpr_spill="$3} END {print n, v, sg, sc, sp}' |
  sed -E 's/_ZN3gpf[0-9]+([a-z_]+)ILi([0-9])ELi([0-9])ELi([0-9])ELi([0-9])ELi([0-9])ELb([01])ELb([01])E[A-Za-z0-9_]* /\1<NB=\2,ST=\3,IPW=\4,MINW=\5,WPI=\6,TC=\7,YR=\8> /'
