#!/bin/bash
# A/B of the IVF chunk length on one rank's share of an 8-way sharded 10M index (needs the -DZVEC_HIP_TUNING variant)
cd $GRAFT_REPO_ROOT
export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_tun.so
for t in 0 2 3 4 6 8; do
  if [ $t = 0 ]; then unset ZVEC_HIP_IVF_TPC; else export ZVEC_HIP_IVF_TPC=$t; fi
  python bench.py --shard-of 8 --steps 40 --warmup 5 --no-host-path "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('tpc=$t', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'fixed %.3f' % r['fixed_ms_per_step'])"
done
