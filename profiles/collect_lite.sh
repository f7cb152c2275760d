#!/bin/bash
# The three passes the bench line depends on (kernel durations, HBM read / write bytes) + the bench line itself; the SQ / TA / TCP
# passes of collect.sh are unchanged by launch-policy changes.  profiles/collect_lite.sh (through gpurun, from the repo root)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o f -- $B --steps 10 --warmup 3 > $R/gpurun_out/prof.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- $B --steps 2 --warmup 1 > $R/gpurun_out/pmc_write.log 2>&1
cd $R && timeout 600 python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench.json
cut -c1-300 gpurun_out/bench.json; find gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" | head
