#!/bin/bash
# Developer tool (GPU box): same-box A/B of library builds.  usage: tools/ab.sh <out.log> <lib name or "new"> ...
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$1; shift
Q="--no-secondary --no-cpu-baseline --no-oracle-check --windows 3"
one() { python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.4g steps/s  %.1f us/launch  conv=%.3f' % (d['value'], d['roofline']['avg_launch_us'], d['frac_converged']))"; }
for lib in "$@"; do
  if [ "$lib" = new ]; then unset GRIDPF_LIB; else export GRIDPF_LIB=$R/grid2op_amd/libgridpf_$lib.so; fi
  for rep in 1 2; do
  echo "[$lib] case14 16/launch obs : $(one --steps 800 --warmup 32)" >> $out
  echo "[$lib] case14 1/launch      : $(one --steps 400 --warmup 32 --steps-per-launch 1)" >> $out
  done
  echo "[$lib] n1 neurips 1024x60   : $(one --env l2rpn_neurips_2020_track1 --batch 1024 --n1 --steps 48 --warmup 16)" >> $out
  echo "[$lib] wcci 1024            : $(one --env l2rpn_wcci_2022_dev --batch 1024 --steps 160 --warmup 16)" >> $out
  echo "[$lib] neurips 4096         : $(one --env l2rpn_neurips_2020_track1 --batch 4096 --steps 160 --warmup 16)" >> $out
done
cat $out
