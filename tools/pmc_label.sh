#!/bin/bash
# usage (on the GPU box): tools/pmc_label.sh <tag>
# PMC passes over tools/time_label.py (each its own run, --kernel-trace only, as gpurun requires) for the labelling kernel
# assign256_f16_kernel / assign_kernel<true> (fp16 rows, 2M x 16384 x 768: the long launches, one chunk of rows each; ZVEC_HIP_ASSIGN256=0 selects
# the latter) and assign_kernel<false> (fp32, 1M x 4096):
# per-launch averages, the MFMA-busy share of the launch and the effective clock (MI355X_MICROARCH.md "DVFS give-back").
# Writes gpurun_out/pmcl_<tag>.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcl_${tag}_$i -- python3 $R/tools/time_label.py > $R/gpurun_out/pmcl_${tag}_$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections,json
res={}
for kern, lo in (("assign256_f16_kernel", 3000.0), ("assign_kernel<true>", 5000.0), ("assign_kernel<false>", 5000.0)):
    out={}
    for i in range(1,5):
        agg=collections.defaultdict(list)
        rows=[]
        for f in glob.glob("$R/gpurun_out/pmcl_${tag}_%d/**/*kernel_trace.csv"%i, recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"]:
                    rows.append((row.get("Dispatch_Id") or row.get("Correlation_Id"), (int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e3))
        big={d for d,t in rows if t>lo}
        durs=[t for d,t in rows if t>lo]
        for f in glob.glob("$R/gpurun_out/pmcl_${tag}_%d/**/*counter_collection.csv"%i, recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"] and (row.get("Dispatch_Id") in big):
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,v in sorted(agg.items()):
            out[k]=sum(v)/max(1,len(v))
        if durs: out["duration_us_pass%d"%i]=sum(durs)/len(durs); out["launches_pass%d"%i]=len(durs)
    if "GRBM_GUI_ACTIVE" in out and "duration_us_pass4" in out:
        out["effective_clock_mhz"]=out["GRBM_GUI_ACTIVE"]/8.0/out["duration_us_pass4"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "duration_us_pass1" in out:
        out["mfma_busy_fraction_at_2400"]=out["SQ_VALU_MFMA_BUSY_CYCLES"]/1024.0/(out["duration_us_pass1"]*2400.0)
    res[kern]=out
json.dump(res,open("$R/gpurun_out/pmcl_${tag}.json","w"),indent=1)
print(json.dumps(res,indent=1))
PY
