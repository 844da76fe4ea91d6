#!/bin/bash
# Developer tool (GPU box): round-6 rocprofv3 profiles of the BASELINE configs as bench.py runs them under the driver's command
# (20 env steps per launch, observation trajectory on in every dispatch: bench.py --profile).
# usage: tools/profile_r06.sh <which: case14|case14_1|n1|n1_118|wcci|idf|ptdf_build|all ...>
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
prof() {   # label, env steps per dispatch, algorithmic bytes per DISPATCH, kernel filter, pmc-set list, stats-pass extra args, bench args...
  label=$1; spl=$2; alg=$3; kf=$4; sets=$5; extra=$6; shift 6
  O=$R/gpurun_out/prof/$label; mkdir -p $O
  # the duration pass runs more launches than the counter passes: an idle MI355X needs tens of milliseconds of work to reach its clocks
  rocprofv3 --kernel-trace --stats -d $O/stats -- python bench.py --profile --no-cpu-baseline --no-oracle-check "$@" $extra > $O/stats.log 2>&1
  dbs=""
  for set in $sets; do
    name=${set%%:*}; ctrs=$(echo ${set#*:} | tr ',' ' ')
    rocprofv3 --kernel-trace --pmc $ctrs -d $O/$name -- python bench.py --profile --no-cpu-baseline --no-oracle-check "$@" > $O/$name.log 2>&1
    dbs="$dbs $(find $O/$name -name '*_results.db' | head -1)"
  done
  python profiles/summarize_rocprof.py "$label: python bench.py --profile --no-cpu-baseline --no-oracle-check $* (1x MI355X; every step-kernel dispatch = $spl env steps, observation trajectory on)" \
      $(find $O/stats -name "*_results.db" | head -1) $dbs > $R/gpurun_out/prof/$label.txt
  python tools/make_traffic_json.py $R/gpurun_out/prof/$label.txt $spl $alg "$kf" > $R/gpurun_out/prof/${label}_traffic.json 2> $O/traffic.err || cat $O/traffic.err
  tail -1 $O/stats.log | cut -c1-300
}
FULL="pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE pmc_sq1:SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS pmc_sq2:SQ_INSTS_SMEM,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_WAVE_CYCLES pmc_sq3:SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY pmc_sq4:SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE pmc_sq5:SQ_INSTS_VALU_FMA_F64,SQ_INSTS_VALU_MUL_F64,SQ_INSTS_VALU_ADD_F64,SQ_INSTS_VALU_TRANS_F64"
LITE="pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE pmc_sq1:SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS pmc_sq3:SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY pmc_sq4:SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE"
MEM="pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE pmc_sq4:SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE"
for w in "$@"; do
  case $w in
    # algorithmic bytes per dispatch = SURVEY 8(d) bytes per env step x lanes of the dispatch x 20 steps
    case14) prof r06_step_kernel_case14 20 $((1472*4096*20)) "step_sparse_kernel<1, 2, 2" "$FULL" "--steps 400 --warmup 400" --steps 60 --warmup 20 --steps-per-launch 20 --windows 2 --no-secondary ;;
    case14_1) prof r06_step_kernel_case14_1perlaunch 1 $((1472*4096)) "step_sparse_kernel<1, 2, 2" "pmc_fetch:FETCH_SIZE pmc_write:WRITE_SIZE" "--steps 400 --warmup 400" --steps 40 --warmup 20 --windows 2 --no-secondary --steps-per-launch 1 ;;
    n1) prof r06_step_kernel_n1_neurips36 20 $((4639*61440*20)) "step_sparse_kernel<1, 0, 1, 2, 1" "$LITE" "" --only n1_fanout --steps 20 --warmup 5 --steps-per-launch 20 ;;
    n1_118) prof r06_step_kernel_n1_wcci118 16 $((13790*191488*16)) "step_sparse_kernel<1, 0, 1, 2, 2" "$LITE" "" --only n1_fanout_118 --steps 20 --warmup 5 ;;
    wcci) prof r06_step_kernel_wcci118 20 $((13790*1024*20)) "step_sparse_kernel<1, 0, 1, 2, 2" "$LITE" "--steps 200 --warmup 40" --only secondary --steps 60 --warmup 20 --steps-per-launch 20 ;;
    # l2rpn_idf_2023, 2 048 lanes: every step launch is two dispatches of 1 024 lanes (one residency round each)
    idf) prof r06_kernels_idf118 20 $((14014*1024*20)) "step_sparse_kernel<1, 0, 1, 2, 2" "$LITE" "--steps 200 --warmup 40" --only dc_ptdf --steps 60 --warmup 20 --steps-per-launch 20 ;;
    ptdf_build) prof r06_ptdf_build_batch 1 956301312 "ptdf_build" "$MEM" "" --only ptdf_build_batch --steps 20 --warmup 5 ;;
    all) O=$R/gpurun_out/prof/r06_all; mkdir -p $O
         rocprofv3 --kernel-trace --stats -d $O/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
         python profiles/summarize_rocprof.py "r06_all_kernels_bench: python bench.py --steps 20 --warmup 5 --no-cpu-baseline (1x MI355X; every kernel of the driver's run)" $(find $O/stats -name "*_results.db" | head -1) > $R/gpurun_out/prof/r06_all_kernels_bench.txt
         tail -1 $O/stats.log | cut -c1-300 ;;
  esac
done
