B=tools/_build
for exe in lu_bench_bar lu_bench; do
for g in l2rpn_case14_sandbox l2rpn_wcci_2022_dev; do
  for ipw in 1 2; do echo "$exe $g ipw=$ipw"; $B/$exe $B/$g.graph 4096 50 $ipw | tail -1;  $B/$exe $B/$g.graph 256 50 $ipw | tail -1; done
done
done
