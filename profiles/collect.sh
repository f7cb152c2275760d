#!/bin/bash
# Collects the evidence the bench line cites, on an MI355X box (run through gpurun from the repo root):  profiles/collect.sh [tag]
#   1. rocprofv3 --kernel-trace --stats of `python bench.py`            -> gpurun_out/prof/f_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE, SQ_*, TA / TCP in SEPARATE passes (never together with a trace domain other than
#      --kernel-trace; every pass under its own `timeout`: a pass that asks for more counters than a block has hangs)
#   3. the bench line itself                                             -> gpurun_out/bench.json
# then profiles/pmc_to_json.py turns (2) into profiles/<tag>_pmc_traffic.json (per-launch HBM bytes, FETCH_SIZE
# doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o f -- $B --steps 10 --warmup 3 > $R/gpurun_out/prof.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_write.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_sq.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_sq2 -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_sq2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum --output-format csv -d $R/gpurun_out/pmc_ta -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_ta.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/pmc_tcp -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_tcp.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_grbm -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_grbm.log 2>&1
cd $R && timeout 600 python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench.json
cut -c1-300 gpurun_out/bench.json; ls gpurun_out/pmc_ta gpurun_out/pmc_tcp 2>&1 | head
