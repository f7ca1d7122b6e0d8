#!/bin/bash
cd $GRAFT_REPO_ROOT
export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_tun.so
for args in "--shard-of 8 --steps 60" "--no-cpu-baseline --steps 20"; do
for hp in 50 75 50 75; do
  export ZVEC_HIP_IVF_HEAD_PCT=$hp
  python bench.py $args --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$args head=$hp', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'])"
done
done
