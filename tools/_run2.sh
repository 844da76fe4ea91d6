B=tools/_build
for g in l2rpn_case14_sandbox l2rpn_neurips_2020_track1 l2rpn_wcci_2022_dev; do
  for ipw in 1 2; do echo "$g ipw=$ipw"; $B/lu_bench $B/$g.graph 4096 50 $ipw | tail -1;  $B/lu_bench $B/$g.graph 256 50 $ipw | tail -1; done
done
