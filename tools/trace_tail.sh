#!/bin/bash
# tools/trace_tail.sh <tag> <bench args...>: kernel trace of bench.py, prints the last dispatches (one timed step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$tag -- python3 $R/bench.py "$@" > $R/gpurun_out/tr_$tag.json 2> $R/gpurun_out/tr_$tag.log || exit 1
python3 - <<PY
import csv,glob,json
d=json.load(open("$R/gpurun_out/tr_$tag.json")); print("$tag", round(d["value"]), d["ms_per_step"])
f=glob.glob("$R/gpurun_out/tr_$tag/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-${TAILN:-18}:]
t0=int(tail[0]["Start_Timestamp"])
for r in tail: print("%-72s %9.1f %8.1f" % (r["Kernel_Name"][:72], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
