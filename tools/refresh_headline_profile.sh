#!/bin/bash
# The headline evidence alone (tools/refresh_profiles.sh does everything): rocprofv3 --kernel-trace --stats of the default bench
# -> gpurun_out/profiles_new/<tag>_ivf10m_{kernel_stats.csv,kernel_trace_tail.csv,bench_under_rocprof.json}
set -e
tag=${1:-r4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
mkdir -p $O && rm -rf $O/stats
# two runs: the default command (two un-gated lanes: a lane's scan is dispatched while the other's still holds the CUs, so every second
# launch "lasts" two scans in the trace) and the same with ONE lane (--streams 1: clean per-launch durations of the list scan)
for v in "" "_streams1"; do
  extra=""; [ -n "$v" ] && extra="--streams 1"
  rm -rf $O/stats
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra > $O/${tag}_ivf10m${v}_bench_under_rocprof.json 2> $O/stats$v.log
  python3 - <<PY
import csv, glob, shutil
O, tag, v = "$O", "$tag", "$v"
def one(pat):
    return sorted(glob.glob(O + "/" + pat, recursive=True))[0]
shutil.copy(one("stats/**/*kernel_stats.csv"), O + "/%s_ivf10m%s_kernel_stats.csv" % (tag, v))
rows = list(csv.DictReader(open(one("stats/**/*kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
scan = [r for r in rows if "scan_kernel<1, true, false, false>" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in scan]
print("list scan launches%s:" % v, len(d), "durations ms:", " ".join("%.2f" % x for x in d))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"]
keep = [k for k in keep if k in rows[0]]
with open(O + "/%s_ivf10m%s_kernel_trace_tail.csv" % (tag, v), "w", newline="") as f:
    w = csv.writer(f); w.writerow(keep + ["Duration_us"])
    for r in rows[-45:]:
        w.writerow([r[k] for k in keep] + ["%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
PY
done
rm -rf $O/stats
