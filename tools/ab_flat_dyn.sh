#!/bin/bash
# A/B of the wide flat kernel's item dealing on one GPU box (needs the -DZVEC_HIP_TUNING variant: tools/build_variant.sh tun)
cd $GRAFT_REPO_ROOT
export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_tun.so
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --workload flat1m --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', 'qps %.0f' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'TF %.1f' % d['roofline']['achieved'], 'fixed_ms %.3f' % d['roofline']['fixed_ms_per_step'])"
}
run static ZVEC_HIP_NO_FLAT_DYN=1
run r2d4 ZVEC_HIP_FLAT_ROUNDS=2 ZVEC_HIP_FLAT_TAIL_DIV=4
run r1d4 ZVEC_HIP_FLAT_ROUNDS=1 ZVEC_HIP_FLAT_TAIL_DIV=4
run r3d4 ZVEC_HIP_FLAT_ROUNDS=3 ZVEC_HIP_FLAT_TAIL_DIV=4
run r2d2 ZVEC_HIP_FLAT_ROUNDS=2 ZVEC_HIP_FLAT_TAIL_DIV=2
run r4d2 ZVEC_HIP_FLAT_ROUNDS=4 ZVEC_HIP_FLAT_TAIL_DIV=2
run r2d8 ZVEC_HIP_FLAT_ROUNDS=2 ZVEC_HIP_FLAT_TAIL_DIV=8
run static2 ZVEC_HIP_NO_FLAT_DYN=1
