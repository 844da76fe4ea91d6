for ipw in 4 2; do
echo "IPW=$ipw case14"; GRIDPF_IPW=$ipw python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-160
done
for ipw in 1 2; do
echo "IPW=$ipw neurips"; GRIDPF_IPW=$ipw python bench.py --env l2rpn_neurips_2020_track1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
done
for ipw in 2 4; do
echo "IPW=$ipw case5"; GRIDPF_IPW=$ipw python bench.py --env rte_case5_example --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
done
