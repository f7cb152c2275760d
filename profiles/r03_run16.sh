#!/bin/bash
# round 3, device run 16: the record for the round (final tree) - whole device suite, rocprofv3 kernel stats and HBM traffic (separate --pmc passes) of the bench
# command, the full bench line (cpu baseline + other configs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03
O=$R/gpurun_out/r03/run16.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 900 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python -m pytest tests -m gpu -x -q
run python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/r03/prof16
rm -rf $P; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o f -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $P/stats.log 2>&1
rc=$?; echo "rocprofv3 stats rc $rc" >> $O
if [ $rc -eq 0 ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $P/pmc_$c.log 2>&1; rc=$?; echo "pmc $c rc $rc" >> $O
    [ $rc -ne 0 ] && break
  done
fi
find $P -name "*_counter_collection.csv" | while read f; do head -1 $f > $f.km; grep "km_" $f >> $f.km; rm $f; done
find $P -name "*_kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
cd $R
run python bench.py --steps 20 --warmup 5
grep "^{" $O | tail -1 > gpurun_out/r03/bench_run16.json
grep -v "^{" $O | grep -v "amdgpu.ids\|^\.\.\." | tail -30
ls -R $P | head -40
