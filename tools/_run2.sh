B=tools/_build
for g in l2rpn_case14_sandbox l2rpn_neurips_2020_track1; do
  for ipw in 1 2 4; do $B/lu_bench $B/$g.graph 4096 50 $ipw | tail -2; done
done
