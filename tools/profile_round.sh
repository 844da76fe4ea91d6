#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel-trace stats + PMC passes of the default bench workload, summarised into
# gpurun_out/prof/<label>.txt by profiles/summarize_rocprof.py.   usage: tools/profile_round.sh <label> [bench args...]
label=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof/$label
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="--steps 160 --warmup 16 --windows 2 --no-cpu-baseline --no-secondary --no-oracle-check $*"
rocprofv3 --kernel-trace --stats -d $O/stats -- python bench.py $ARGS > $O/stats.log 2>&1
for set in "pmc_fetch:FETCH_SIZE" "pmc_write:WRITE_SIZE" "pmc_sq1:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "pmc_sq2:SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "pmc_sq3:SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "pmc_sq4:SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "pmc_sq5:SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
  name=${set%%:*}; ctrs=${set#*:}
  rocprofv3 --kernel-trace --pmc $ctrs -d $O/$name -- python bench.py --steps 48 --warmup 16 --windows 1 --no-cpu-baseline --no-secondary --no-oracle-check $* > $O/$name.log 2>&1
done
python profiles/summarize_rocprof.py "$label: python bench.py $ARGS (1x MI355X)" $(find $O/stats -name "*_results.db" | head -1) $(for n in pmc_fetch pmc_write pmc_sq1 pmc_sq2 pmc_sq3 pmc_sq4 pmc_sq5; do find $O/$n -name "*_results.db" | head -1; done) > $R/gpurun_out/prof/$label.txt
tail -1 $O/stats.log | cut -c1-300
