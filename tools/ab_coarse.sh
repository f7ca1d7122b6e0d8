#!/bin/bash
cd $GRAFT_REPO_ROOT
export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_tun.so
for v in wide narrow wide narrow; do
  if [ $v = wide ]; then unset ZVEC_HIP_NO_WIDE_DUMP; else export ZVEC_HIP_NO_WIDE_DUMP=1; fi
  python bench.py --shard-of 8 --steps 40 --warmup 5 --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'fixed %.4f' % r['fixed_ms_per_step'])"
done
