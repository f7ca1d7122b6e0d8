#!/bin/bash
# usage (on the GPU box): tools/refresh_flat16_profile.sh [tag]
# The evidence of the wide fp16 flat scan (bench.py --workload flat1m_fp16: 1M x 768 fp16, k = 10), for 256 and 1024 queries:
#   <tag>_flat1m_fp16[_b1024]_bench.json                 the plain bench line (256 x 256 tile; two lanes, the default)
#   <tag>_flat1m_fp16[_b1024]_streams1_bench.json        the same on ONE lane (--streams 1: kernel_ms is then one launch alone)
#   <tag>_flat1m_fp16[_b1024]_scan8_bench.json           one lane with option scan256 = 0 (the 128 x 128 tile of rounds 2-3)
#   <tag>_flat1m_fp16[_b1024]_kernel_stats.csv / _kernel_trace_tail.csv / _bench_under_rocprof.json   rocprofv3 --kernel-trace --stats, one lane
#   <tag>_flat1m_fp16_pmc.json                            MFMA-busy share / waits / LDS conflicts / effective clock of scan256_f16_kernel
#   <tag>_flat1m_fp16_traffic.json                        its HBM-side bytes per launch (FETCH_SIZE / WRITE_SIZE passes) against the algorithmic bytes
# -> gpurun_out/profiles_new/
set -e
tag=${1:-r4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
mkdir -p $O
for b in 256 1024; do
  v=""; [ $b != 256 ] && v="_b$b"
  python3 $R/bench.py --workload flat1m_fp16 --batch $b --steps 30 --warmup 5 --no-cpu-baseline > $O/${tag}_flat1m_fp16${v}_bench.json 2> $O/f16$v.log
  python3 $R/bench.py --workload flat1m_fp16 --batch $b --steps 30 --warmup 5 --no-cpu-baseline --streams 1 > $O/${tag}_flat1m_fp16${v}_streams1_bench.json 2>> $O/f16$v.log
  ZVEC_HIP_SCAN256=0 python3 $R/bench.py --workload flat1m_fp16 --batch $b --steps 30 --warmup 5 --no-cpu-baseline --streams 1 > $O/${tag}_flat1m_fp16${v}_scan8_bench.json 2>> $O/f16$v.log
  rm -rf $O/f16prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f16prof -- python3 $R/bench.py --workload flat1m_fp16 --batch $b --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/${tag}_flat1m_fp16${v}_bench_under_rocprof.json 2>> $O/f16$v.log
  python3 - <<PY
import csv, glob, shutil
O, tag, v = "$O", "$tag", "$v"
def one(pat):
    return sorted(glob.glob(O + "/" + pat, recursive=True))[0]
shutil.copy(one("f16prof/**/*kernel_stats.csv"), O + "/%s_flat1m_fp16%s_kernel_stats.csv" % (tag, v))
rows = list(csv.DictReader(open(one("f16prof/**/*kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"]
keep = [k for k in keep if k in rows[0]]
with open(O + "/%s_flat1m_fp16%s_kernel_trace_tail.csv" % (tag, v), "w", newline="") as f:
    w = csv.writer(f); w.writerow(keep + ["Duration_us"])
    for r in rows[-45:]:
        w.writerow([r[k][:120] for k in keep] + ["%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
PY
  rm -rf $O/f16prof
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  rm -rf $O/f16pmc_$i
  timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/f16pmc_$i -- python3 $R/bench.py --workload flat1m_fp16 --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/f16pmc_$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections, json
O, tag = "$O", "$tag"
kern = "scan256_f16_kernel"
out = {}
for i in range(1, 5):
    agg = collections.defaultdict(list)
    durs = []
    for f in glob.glob(O + "/f16pmc_%d/**/*kernel_trace.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for f in glob.glob(O + "/f16pmc_%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(agg.items()):
        out[k] = sum(v) / max(1, len(v))
    if durs:
        out["duration_us_pass%d" % i] = sum(durs) / len(durs)
        out["launches_pass%d" % i] = len(durs)
if "GRBM_GUI_ACTIVE" in out and "duration_us_pass4" in out:
    out["effective_clock_mhz"] = out["GRBM_GUI_ACTIVE"] / 8.0 / out["duration_us_pass4"]
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "duration_us_pass1" in out:
    out["mfma_busy_fraction_at_2400"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (out["duration_us_pass1"] * 2400.0)
json.dump({kern: out}, open(O + "/%s_flat1m_fp16_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/f16pmc_*
# HBM-side traffic of the scan kernel (MI355X_MICROARCH.md, HBM section: separate passes, FETCH_SIZE doubled on gfx950 for 16 B / lane reads)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/f16t_$c
  timeout -k 10 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/f16t_$c -- python3 $R/bench.py --workload flat1m_fp16 --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $O/f16t_$c.json 2> $O/f16t_$c.log || exit 1
done
python3 - <<PY
import csv, glob, json
O, tag = "$O", "$tag"
def per_launch(c):
    vals = []
    for f in glob.glob(O + "/f16t_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "scan256_f16_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                vals.append(float(row["Counter_Value"]))
    vals = vals[-10:]
    return sum(vals) / len(vals), len(vals)
fetch, nf = per_launch("FETCH_SIZE")
write, nw = per_launch("WRITE_SIZE")
b = json.loads(open(O + "/f16t_FETCH_SIZE.json").read().strip().splitlines()[-1])
alg = b["roofline"]["algorithmic_bytes"]
traffic = fetch * 1024 * 2 + write * 1024
json.dump({"kernel": "zvk::scan256_f16_kernel - wide flat scan of fp16 rows (base tiles and query rows by LDS-DMA, 16 B per lane)",
           "workload": b["config"]["workload"], "launches_averaged": nf,
           "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
           "correction": "bytes = FETCH_SIZE*1024*2 (gfx950 tallies the 128-B requests of 16 B/lane reads at 64 B: MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
           "hbm_traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg},
          open(O + "/%s_flat1m_fp16_traffic.json" % tag, "w"), indent=1)
print("flat1m_fp16 traffic/algorithmic", traffic / alg, "fetch KiB", fetch, "write KiB", write)
PY
rm -rf $O/f16t_*
