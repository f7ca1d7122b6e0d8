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
# HBM traffic of the single-query scoring kernel: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/b1f -- python3 $R/bench.py --batch 1 --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-host-path > $O/b1_pmc_fetch.json 2> $O/b1f.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/b1w -- python3 $R/bench.py --batch 1 --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-host-path > $O/b1_pmc_write.json 2> $O/b1w.log
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
import json
def per_launch(d, counter, kern):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(one(d + "/**/*counter_collection.csv")))
            if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
    vals = vals[-20:]           # the timed launches
    return sum(vals) / len(vals), len(vals)
fetch, nf = per_launch("b1f", "FETCH_SIZE", "pkeys_topk_kernel")
write, nw = per_launch("b1w", "WRITE_SIZE", "pkeys_topk_kernel")
b = json.load(open(O + "/b1_pmc_fetch.json"))
alg = b["roofline"]["algorithmic_bytes"]
traffic = fetch * 1024 * 2 + write * 1024
json.dump({"kernel": "zvk::pkeys_topk_kernel<false> - the single-query IVF route's scoring kernel (a wave per four probed rows, 16-byte chunk loads)",
           "workload": b["config"]["workload"], "launches_averaged": nf,
           "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
           "correction": "bytes = FETCH_SIZE*1024*2 (gfx950 tallies the 128-B requests of 16 B/lane reads at 64 B: MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
           "hbm_traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg},
          open(O + "/%s_ivf10m_b1_pmc.json" % tag, "w"), indent=1)
print("b1 traffic/algorithmic", traffic / alg, "-> add \"ivf10m_b1\": %r to profiles/pmc_traffic.json" % traffic)
PY
rm -rf $O/b1 $O/b8 $O/b1f $O/b1w
ls -la $O
