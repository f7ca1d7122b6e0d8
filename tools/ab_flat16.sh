#!/bin/bash
# usage (GPU box): tools/ab_flat16.sh <variant>[@ENV=VALUE] ...   ("base" = the shipped library, "old" = base with scan256 off):
# the kernel time and QPS of bench.py --workload flat1m_fp16 (BENCH_ARGS adds arguments), twice each.  Variants are diagnostic
# builds of libzvec_hip (zvec_amd/_variants/libzvec_hip_<variant>.so: tools/build_variant.sh <variant> -DZVK_S256_...).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for spec in "$@"; do
  v=${spec%%@*}
  unset ZVEC_HIP_LIBRARY ZVEC_HIP_SCAN256 ZVEC_HIP_SEED_ROWS256
  if [ "$spec" != "$v" ]; then export "${spec#*@}"; fi
  if [ $v = old ]; then export ZVEC_HIP_SCAN256=0; elif [ $v != base ]; then export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_$v.so; fi
  echo -n "$spec: "; python bench.py --workload flat1m_fp16 --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.0f QPS  step %.3f ms  kernel %.3f ms  %.0f TF  fixed %.3f ms  recall %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['achieved'], r.get('fixed_ms_per_step') or 0, d['config'].get('recall_at_10')))"
done
done
