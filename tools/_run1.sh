python bench.py --env l2rpn_neurips_2020_track1 --steps 50 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
GRIDPF_IPW=2 python bench.py --env l2rpn_neurips_2020_track1 --steps 50 --warmup 10 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
python bench.py --env l2rpn_neurips_2020_track1 --n1 --batch 1024 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
