#!/bin/bash
# HBM-side traffic of the IVF list scan over the fp16 shadow lists (bench.py's certified_half_scan leg, one lane):
# separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md, HBM section; FETCH_SIZE doubled on gfx950 for 16 B / lane reads)
# -> gpurun_out/profiles_new/<tag>_ivf10m_shadow_traffic.json
set -e
tag=${1:-r4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/sht_$c
  timeout -k 10 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/sht_$c -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-path --streams 1 > $O/sht_$c.json 2> $O/sht_$c.log || exit 1
done
python3 - <<PY
import csv, glob, json
O, tag = "$O", "$tag"
def per_launch(c):
    vals = []
    for f in glob.glob(O + "/sht_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "scan_kernel<1, true, false, true>" in row["Kernel_Name"] and row["Counter_Name"] == c:
                vals.append(float(row["Counter_Value"]))
    vals = vals[-6:]
    return sum(vals) / len(vals), len(vals)
fetch, nf = per_launch("FETCH_SIZE")
write, nw = per_launch("WRITE_SIZE")
b = json.loads(open(O + "/sht_FETCH_SIZE.json").read().strip().splitlines()[-1])
alg = b["certified_half_scan"]["algorithmic_bytes"]
traffic = fetch * 1024 * 2 + write * 1024
json.dump({"kernel": "zvk::scan_kernel<1, true, false, true> - IVF list scan over the fp16 shadow lists (certified half-width pre-selection)",
           "workload": b["config"]["workload"], "launches_averaged": nf,
           "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
           "correction": "bytes = FETCH_SIZE*1024*2 (gfx950 tallies the 128-B requests of 16 B/lane reads at 64 B: MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
           "hbm_traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg},
          open(O + "/%s_ivf10m_shadow_traffic.json" % tag, "w"), indent=1)
print("ivf10m shadow scan traffic/algorithmic", traffic / alg, "fetch KiB", fetch, "write KiB", write)
PY
rm -rf $O/sht_*
