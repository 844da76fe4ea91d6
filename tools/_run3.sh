cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd $R
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/p$i.log 2>&1
  db=$(find $O/p$i -name "*_results.db" | head -1)
  python - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
for kn, cn, v, n, d in c.execute("select kernel_name,counter_name,avg(value),count(*),avg(duration) from counters_collection where kernel_name like '%step_sparse%' group by kernel_name,counter_name"):
    print(f"{cn:>28} = {v:>18.1f}  (n={n}, avg {d/1000:.1f} us)")
PY
done
