#!/bin/bash
cd $GRAFT_REPO_ROOT
export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_tun.so
for args in "--shard-of 8 --steps 60" "--no-cpu-baseline --steps 20"; do
for cap in 0 2 1; do
  export ZVEC_HIP_IVF_OCC_CAP=$cap
  for st in 2 1; do
  python bench.py $args --streams $st --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$args cap=$cap streams=$st', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'])"
  done
done
done
