B=tools/_build
for g in l2rpn_case14_sandbox l2rpn_wcci_2022_dev; do
  for blocks in 256 4096; do $B/lu_bench $B/$g.graph $blocks 50 | tail -1; done
done
