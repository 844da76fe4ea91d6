python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-160
python bench.py --env l2rpn_neurips_2020_track1 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
python bench.py --env rte_case5_example --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
python bench.py --env l2rpn_wcci_2022_dev --batch 1024 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
