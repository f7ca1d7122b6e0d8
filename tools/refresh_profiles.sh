#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/profiles_new/ (copy what is to be judged into profiles/):
#   1. --kernel-trace --stats of the default bench (ivf10m)           2./3. separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   4. --kernel-trace --stats of the flat1m bench
# Usage on the GPU box: tools/refresh_profiles.sh <round-tag>
set -e
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
rm -rf $O && mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/${tag}_ivf10m_bench_under_rocprof.json 2> $O/stats.log
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/flat -- python3 $R/bench.py --workload flat1m --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/${tag}_flat1m_bench_under_rocprof.json 2> $O/flat.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard8 -- python3 $R/bench.py --shard-of 8 --steps 10 --warmup 2 --no-host-path > $O/${tag}_ivf10m_shard8_bench_under_rocprof.json 2> $O/shard8.log
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s100 -- python3 $R/bench.py --workload ivf100m_fp16 --shard-of 8 --steps 10 --warmup 2 --no-host-path > $O/${tag}_ivf100m_fp16_shard8_bench_under_rocprof.json 2> $O/s100.log
# round 3: the product's count = 1 / small-batch route, and the build's labelling kernel
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -- python3 $R/bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-host-path > $O/${tag}_ivf10m_b1_bench_under_rocprof.json 2> $O/b1.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b8 -- python3 $R/bench.py --batch 8 --steps 100 --warmup 10 --no-cpu-baseline --no-host-path > $O/${tag}_ivf10m_b8_bench_under_rocprof.json 2> $O/b8.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/label -- python3 $R/tools/time_label.py > $O/${tag}_label_time.log 2> $O/label.log
# round 4: the wide flat tile on fp16 rows (its fraction of the f16 peak must be recomputable from a committed summary)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/flat16 -- python3 $R/bench.py --workload flat1m_fp16 --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/${tag}_flat1m_fp16_bench_under_rocprof.json 2> $O/flat16.log
python3 - <<PY
import csv, glob, json, shutil
O, tag = "$O", "$tag"
def one(pat):
    return sorted(glob.glob(O + "/" + pat, recursive=True))[0]
shutil.copy(one("stats/**/*kernel_stats.csv"), O + "/%s_ivf10m_kernel_stats.csv" % tag)
shutil.copy(one("flat/**/*kernel_stats.csv"), O + "/%s_flat1m_kernel_stats.csv" % tag)
shutil.copy(one("shard8/**/*kernel_stats.csv"), O + "/%s_ivf10m_shard8_kernel_stats.csv" % tag)
shutil.copy(one("s100/**/*kernel_stats.csv"), O + "/%s_ivf100m_fp16_shard8_kernel_stats.csv" % tag)
shutil.copy(one("b1/**/*kernel_stats.csv"), O + "/%s_ivf10m_b1_kernel_stats.csv" % tag)
shutil.copy(one("b8/**/*kernel_stats.csv"), O + "/%s_ivf10m_b8_kernel_stats.csv" % tag)
shutil.copy(one("label/**/*kernel_stats.csv"), O + "/%s_label_kernel_stats.csv" % tag)
shutil.copy(one("flat16/**/*kernel_stats.csv"), O + "/%s_flat1m_fp16_kernel_stats.csv" % tag)
for name, d in (("ivf10m", "stats"), ("flat1m", "flat"), ("ivf10m_shard8", "shard8"), ("ivf10m_b1", "b1"), ("ivf10m_b8", "b8")):
    rows = list(csv.DictReader(open(one(d + "/**/*kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"]
    keep = [k for k in keep if k in rows[0]]
    with open(O + "/%s_%s_kernel_trace_tail.csv" % (tag, name), "w", newline="") as f:
        w = csv.writer(f); w.writerow(keep + ["Duration_us"])
        for r in rows[-45:]:
            w.writerow([r[k] for k in keep] + ["%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
def per_launch(d, counter):
    vals = []
    for row in csv.DictReader(open(one(d + "/**/*counter_collection.csv"))):
        if "scan_kernel<1, true" in row["Kernel_Name"] and row["Counter_Name"] == counter:
            vals.append(float(row["Counter_Value"]))
    vals = vals[-3:]           # the timed launches
    return sum(vals) / len(vals), len(vals)
fetch, nf = per_launch("pmc_fetch", "FETCH_SIZE")
write, nw = per_launch("pmc_write", "WRITE_SIZE")
b = json.load(open(O + "/pmc_fetch.json"))
alg = b["roofline"]["algorithmic_bytes"]
traffic = fetch * 1024 * 2 + write * 1024
json.dump({"kernel": "zvk::scan_kernel<1, true, false, false> - IVF list scan (16x16x4 fp32 MFMA, two 16-row halves, non-temporal base loads)",
           "workload": b["config"]["workload"], "launches_averaged": nf,
           "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
           "correction": "bytes = FETCH_SIZE*1024*2 (gfx950 tallies the 128-B requests of 16 B/lane streaming reads at 64 B: MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
           "hbm_traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg},
          open(O + "/%s_ivf10m_pmc.json" % tag, "w"), indent=1)
json.dump({"ivf10m": traffic}, open(O + "/pmc_traffic.json", "w"))
print("traffic/algorithmic", traffic / alg)
PY
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/flat $O/shard8 $O/s100 $O/b1 $O/b8 $O/label
ls -la $O
