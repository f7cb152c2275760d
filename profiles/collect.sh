#!/bin/bash
# Collects the evidence the bench line cites, on an MI355X box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py`            -> gpurun_out/prof/f_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE, SQ_* in SEPARATE passes  -> gpurun_out/pmc_{fetch,write,sq}/
#   3. the bench line itself                                             -> gpurun_out/bench.json
# then profiles/pmc_to_json.py turns (2) into profiles/<tag>_pmc_traffic.json (per-launch HBM bytes, FETCH_SIZE
# doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o f -- $B --steps 10 --warmup 3 > $R/gpurun_out/prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_sq.log 2>&1
cd $R && python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench.json
cut -c1-300 gpurun_out/bench.json
