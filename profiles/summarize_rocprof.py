#!/usr/bin/env python
"""Turn rocprofv3 rocpd sqlite outputs (``*_results.db``) into the small text summaries committed under
profiles/.  Usage: summarize_rocprof.py <label> <stats.db> [<pmc.db> ...] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    label, stats_db, pmc_dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    print(f"# {label}")
    cur = sqlite3.connect(stats_db).cursor()
    print("\n## rocprofv3 --kernel-trace --stats : per-kernel summary (durations in us)")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{calls:>6} {total:>12.3f} {avg:>10.3f} {pct:>7.2f}  {name}")
    print("\n## dispatch geometry / resources of the solver kernels")
    q = ("select name,count(*),avg(duration),min(duration),max(duration),max(grid_x),max(workgroup_x),max(lds_size),"
         "max(vgpr_count),max(accum_vgpr_count),max(sgpr_count),max(scratch_size) from kernels where name like '%gpf::%' group by name")
    for r in cur.execute(q):
        print(f"{r[0]}\n   dispatches={r[1]} avg_ns={r[2]:.0f} min_ns={r[3]} max_ns={r[4]} grid_x={r[5]} wg_x={r[6]} "
              f"lds_bytes={r[7]} vgpr={r[8]} agpr={r[9]} sgpr={r[10]} scratch={r[11]}")
    for db in pmc_dbs:
        c = sqlite3.connect(db).cursor()
        print(f"\n## rocprofv3 --pmc ({db.split('/')[-2]}): average counter value per dispatch (solver kernels)")
        q = ("select kernel_name,counter_name,avg(value),count(*),avg(duration) from counters_collection "
             "where kernel_name like '%gpf::%' group by kernel_name,counter_name")
        for kn, cn, v, n, d in c.execute(q):
            print(f"{cn:>28} = {v:>16.1f}   (n={n}, avg dispatch {d/1000:.1f} us)  {kn.split('(')[0]}")


if __name__ == "__main__":
    main()
