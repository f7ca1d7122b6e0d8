#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <workload> <variant> [<variant> ...]   ("base" = the shipped library); each run twice
cd $GRAFT_REPO_ROOT
wl=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset ZVEC_HIP_LIBRARY; else export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_$v.so; fi
  python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v', 'qps %.0f' % d['value'], 'kernel_ms %.4f' % r['kernel_ms'], 'achieved %.1f %s' % (r['achieved'], r['unit']), 'fixed_ms %.3f' % r['fixed_ms_per_step'])"
done
done
