#!/bin/bash
# The small-batch part of tools/refresh_profiles.sh alone (the product's count = 1 calls): kernel trace + stats of
# `bench.py --batch 1` and `--batch 8` under gpurun_out/profiles_new/.   Usage on the GPU box: tools/refresh_small_batch_profiles.sh <round-tag>
set -e
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
rm -rf $O && mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -- python3 $R/bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-host-path > $O/${tag}_ivf10m_b1_bench_under_rocprof.json 2> $O/b1.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b8 -- python3 $R/bench.py --batch 8 --steps 100 --warmup 10 --no-cpu-baseline --no-host-path > $O/${tag}_ivf10m_b8_bench_under_rocprof.json 2> $O/b8.log
python3 - <<PY
import csv, glob, shutil
O, tag = "$O", "$tag"
def one(pat):
    return sorted(glob.glob(O + "/" + pat, recursive=True))[0]
for name, d in (("ivf10m_b1", "b1"), ("ivf10m_b8", "b8")):
    shutil.copy(one(d + "/**/*kernel_stats.csv"), O + "/%s_%s_kernel_stats.csv" % (tag, name))
    rows = list(csv.DictReader(open(one(d + "/**/*kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"]
    keep = [k for k in keep if k in rows[0]]
    with open(O + "/%s_%s_kernel_trace_tail.csv" % (tag, name), "w", newline="") as f:
        w = csv.writer(f); w.writerow(keep + ["Duration_us"])
        for r in rows[-45:]:
            w.writerow([r[k] for k in keep] + ["%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
PY
rm -rf $O/b1 $O/b8
ls -la $O
