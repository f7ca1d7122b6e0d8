#!/bin/bash
# usage (on the GPU box): tools/pmc_flat.sh <tag> [extra bench args]
# PMC passes over `bench.py --workload flat1m` (each its own run, --kernel-trace only, as gpurun requires) for the wide
# flat kernel scan8_kernel; prints per-launch averages and the effective clock GRBM_GUI_ACTIVE / 8 / wall
# (MI355X_MICROARCH.md "DVFS give-back").  Writes gpurun_out/pmcf_<tag>.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_${tag}_$i -- python3 $R/bench.py --workload flat1m --steps 6 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/pmcf_${tag}_$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections,json
out={}
for i in range(1,5):
    agg=collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/pmcf_${tag}_%d/**/*counter_collection.csv"%i, recursive=True):
        for row in csv.DictReader(open(f)):
            if "scan8_kernel" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur=[]
    for f in glob.glob("$R/gpurun_out/pmcf_${tag}_%d/**/*kernel_trace.csv"%i, recursive=True):
        for row in csv.DictReader(open(f)):
            if "scan8_kernel" in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e3)
    big=[d for d in dur if d>500]           # the timed full-size launches (the seed pre-pass is tiny)
    n=len(big)
    for k,v in sorted(agg.items()):
        vv=sorted(v)[-n:] if n else v
        out[k]=sum(vv)/max(1,len(vv))
    if big: out["duration_us_pass%d"%i]=sum(big)/n
if "GRBM_GUI_ACTIVE" in out and "duration_us_pass4" in out:
    out["effective_clock_mhz"]=out["GRBM_GUI_ACTIVE"]/8.0/out["duration_us_pass4"]
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "duration_us_pass1" in out:
    # busy cycles summed over SIMDs of all CUs: 1024 SIMDs
    out["mfma_busy_fraction_at_2400"]=out["SQ_VALU_MFMA_BUSY_CYCLES"]/1024.0/(out["duration_us_pass1"]*2400.0)
json.dump(out,open("$R/gpurun_out/pmcf_${tag}.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
rm -rf $R/gpurun_out/pmcf_${tag}_[0-9]
