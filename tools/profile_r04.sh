#!/bin/bash
# Developer tool (GPU box): round-4 rocprofv3 profiles of the four BASELINE configs, every dispatch of the step kernel a full
# 16-step launch WITH the observation trajectory (bench.py --profile).  usage: tools/profile_r04.sh <which: case14|n1|wcci|idf ...>
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
STATS_EXTRA="--steps 480 --warmup 480"     # (argparse: the last occurrence wins)
prof() {   # label, algorithmic bytes per launch, kernel filter, pmc-set list, bench args...
  label=$1; alg=$2; kf=$3; sets=$4; shift 4
  O=$R/gpurun_out/prof/$label; mkdir -p $O
  # the duration pass runs 10x the launches of the counter passes: an idle MI355X needs tens of milliseconds of work to reach its clocks
  rocprofv3 --kernel-trace --stats -d $O/stats -- python bench.py --profile --no-cpu-baseline --no-oracle-check "$@" $STATS_EXTRA > $O/stats.log 2>&1
  dbs=""
  for set in $sets; do
    name=${set%%:*}; ctrs=$(echo ${set#*:} | tr ',' ' ')
    rocprofv3 --kernel-trace --pmc $ctrs -d $O/$name -- python bench.py --profile --no-cpu-baseline --no-oracle-check "$@" > $O/$name.log 2>&1
    dbs="$dbs $(find $O/$name -name '*_results.db' | head -1)"
  done
  python profiles/summarize_rocprof.py "$label: python bench.py --profile --no-cpu-baseline --no-oracle-check $* (1x MI355X; every step-kernel dispatch = 16 env steps, observation trajectory on)" \
      $(find $O/stats -name "*_results.db" | head -1) $dbs > $R/gpurun_out/prof/$label.txt
  python tools/make_traffic_json.py $R/gpurun_out/prof/$label.txt ${SPL:-16} $alg "$kf" > $R/gpurun_out/prof/${label}_traffic.json 2> $O/traffic.err || cat $O/traffic.err
  tail -1 $O/stats.log | cut -c1-200
}
FULL="pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE pmc_sq1:SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS pmc_sq2:SQ_INSTS_SMEM,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_WAVE_CYCLES pmc_sq3:SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY pmc_sq4:SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE pmc_sq5:SQ_INSTS_VALU_FMA_F64,SQ_INSTS_VALU_MUL_F64,SQ_INSTS_VALU_ADD_F64,SQ_INSTS_VALU_TRANS_F64"
LITE="pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE pmc_sq1:SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS pmc_sq3:SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY pmc_sq4:SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE"
for w in "$@"; do
  case $w in
    case14) prof r04_step_kernel_case14 96468992 "step_sparse_kernel<1, 2, 2" "$FULL" --steps 48 --warmup 16 --windows 2 --no-secondary ;;
    case14_shipped) prof r04_step_kernel_case14_shipped 96468992 "step_sparse_kernel<1, 2, 2" "pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE pmc_sq1:SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS" --steps 48 --warmup 16 --windows 2 --no-secondary --no-jit ;;
    all) O=$R/gpurun_out/prof/r04_all; mkdir -p $O
         rocprofv3 --kernel-trace --stats -d $O/stats -- python bench.py --no-cpu-baseline > $O/stats.log 2>&1
         python profiles/summarize_rocprof.py "r04_all_kernels_bench: python bench.py --no-cpu-baseline (1x MI355X; every kernel of the default run, step kernels specialised at run time)" $(find $O/stats -name "*_results.db" | head -1) > $R/gpurun_out/prof/r04_all_kernels_bench.txt
         tail -1 $O/stats.log | cut -c1-200 ;;
    case14_1) SPL=1 prof r04_step_kernel_case14_1perlaunch 6029312 "step_sparse_kernel<1, 2, 2" "pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE" --steps 48 --warmup 16 --windows 2 --no-secondary --steps-per-launch 1 ;;
    n1) prof r04_step_kernel_n1_neurips36 4560322560 "step_sparse_kernel<1, 0, 1, 2, 1" "$LITE" --only n1_fanout --steps 48 --warmup 16 ;;
    wcci) prof r04_step_kernel_wcci118 225935360 "step_sparse_kernel<1, 0, 1, 2, 2" "$LITE" --only secondary --steps 64 --warmup 16 ;;
    idf) prof r04_kernels_idf118 459210752 "step_sparse_kernel<1, 0, 1, 2, 2" "$LITE" --only dc_ptdf --steps 48 --warmup 16 ;;
  esac
done
