B=tools/_build
for g in l2rpn_case14_sandbox l2rpn_wcci_2022_dev; do for ipw in 1 2; do $B/lu_bench $B/$g.graph 256 50 $ipw | tail -1 | cut -c1-200; done; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-160
python bench.py --env l2rpn_neurips_2020_track1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
python bench.py --env rte_case5_example --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
python bench.py --env l2rpn_wcci_2022_dev --batch 1024 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
