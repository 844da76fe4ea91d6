#!/bin/bash
# Developer tool (GPU box): same-box A/B of kernel experiments selected by JIT flags / environment.
#   tools/ab6.sh <out.log> name1 'ENV=.. ENV2=..' name2 '...'      (an empty env string = the shipped / ahead-of-time kernels)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
out=$1; shift
Q="--no-secondary --no-cpu-baseline --windows 3"
one() { python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.4g steps/s  %.1f us/launch  conv=%.3f oracle=%s' % (d['value'], d['roofline']['avg_launch_us'], d.get('frac_converged', -1), d.get('oracle_ok', d.get('oracle', {}).get('ok', '?'))))"; }
while [ $# -gt 0 ]; do
  name=$1; envs=$2; shift 2
  for rep in 1 2; do
    echo "[$name] wcci 1024 : $(env $envs bash -c "$(declare -f one); Q='$Q'; one --env l2rpn_wcci_2022_dev --batch 1024 --steps 160 --warmup 16")" >> $out
  done
  echo "[$name] idf 2048  : $(env $envs bash -c "$(declare -f one); Q='$Q'; one --env l2rpn_idf_2023 --batch 2048 --steps 160 --warmup 16")" >> $out
done
cat $out
