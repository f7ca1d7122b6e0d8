#!/bin/bash
# A/B of library variants on ONE box: tools/ab_flat.sh <workload> <variant|default|env:VAR=VAL> ...
# prints QPS / roofline.achieved / kernel_ms per arm, two rounds (ABAB) to expose drift
W=$1; shift
mkdir -p gpurun_out
for round in 1 2; do
for v in "$@"; do
  unset ZVEC_HIP_LIBRARY; extra=""
  case $v in
    default) ;;
    env:*) extra="${v#env:}";;
    *) export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_$v.so;;
  esac
  env $extra timeout -k 10 300 python bench.py --workload $W --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline $BENCH_EXTRA > gpurun_out/ab.json 2> gpurun_out/ab.log || { echo "$v FAILED"; tail -3 gpurun_out/ab.log; continue; }
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('%-28s'%'$v', round(d['value']), round(d['roofline']['achieved'],2), round(d['roofline']['kernel_ms'],4), d.get('recall'))"
done; done
