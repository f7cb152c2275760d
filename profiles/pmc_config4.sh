#!/bin/bash
# PMC passes for BASELINE config 4's bicubic warp (run through gpurun from the repo root): profiles/pmc_config4.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
B="python $R/profiles/time_config4.py"
export CFG4_ONLY=1 CFG4_PMC=1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc4_sq -o p -- $B > $R/gpurun_out/pmc4_sq.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc4_sq2 -o p -- $B > $R/gpurun_out/pmc4_sq2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc4_sq3 -o p -- $B > $R/gpurun_out/pmc4_sq3.log 2>&1
python - <<'P'
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc4_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "cubic" not in k and "km_warp_fwd_kernel" not in k: continue
        acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
P
