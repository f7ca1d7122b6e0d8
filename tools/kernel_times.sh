#!/bin/bash
# tools/kernel_times.sh <tag> <bench args...>: average duration of each kernel over the last N dispatches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$tag -- python3 $R/bench.py "$@" > $R/gpurun_out/kt_$tag.json 2> $R/gpurun_out/kt_$tag.log || exit 1
python3 - <<PY
import csv,glob,json,collections
d=json.load(open("$R/gpurun_out/kt_$tag.json")); print("$tag", round(d["value"]), round(d["ms_per_step"],4))
f=glob.glob("$R/gpurun_out/kt_$tag/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-110:]
agg=collections.OrderedDict()
for r in tail:
    agg.setdefault(r["Kernel_Name"][:60],[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in agg.items(): print("   %-62s n=%3d avg %8.1f us" % (k,len(v),sum(v)/len(v)))
PY
