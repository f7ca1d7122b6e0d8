#!/bin/bash
# usage (GPU box): tools/ab_label.sh <variant> [<variant> ...]   ("base" = the shipped library): the f16 labelling figure of tools/time_label.py, twice each
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset ZVEC_HIP_LIBRARY; else export ZVEC_HIP_LIBRARY=$PWD/zvec_amd/_variants/libzvec_hip_$v.so; fi
  echo -n "$v: "; python tools/time_label.py fp16 2>/dev/null | tail -1
done
done
