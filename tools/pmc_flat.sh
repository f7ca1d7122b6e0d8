#!/bin/bash
# usage: tools_pmc_flat.sh <tag> ; runs PMC passes on bench flat1m and extracts scan_kernel rows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$1_$i -- python3 $R/bench.py --workload flat1m --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcf_$1_$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections
for i in range(1,6):
    fs=glob.glob("$R/gpurun_out/pmcf_$1_%d/**/*counter_collection.csv"%i, recursive=True)
    agg=collections.defaultdict(list)
    for f in fs:
        for row in csv.DictReader(open(f)):
            if "scan_kernel" in row["Kernel_Name"] and "Li1ELb0" not in row["Kernel_Name"]:
                agg[(row["Kernel_Name"][:60],row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(k[0],k[1],len(v),sum(v[-2:])/max(1,len(v[-2:])))
PY
