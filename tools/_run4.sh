cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lu
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d $O/a -- tools/_build/lu_bench tools/_build/l2rpn_case14_sandbox.graph 4096 50 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $O/b -- tools/_build/lu_bench tools/_build/l2rpn_case14_sandbox.graph 4096 50 > $O/b.log 2>&1
for d in a b; do
db=$(find $O/$d -name "*_results.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
for did, cn, v, d in c.execute("select dispatch_id,counter_name,value,duration from counters_collection order by dispatch_id,counter_name"):
    print(f"disp {did} {cn:>24} = {v:>16.1f}  ({d/1000:.1f} us)")
PY
done
