#!/bin/bash
# One parametrised device script (replaces the per-run scripts of round 3).  Usage on the GPU box (through gpurun):
#   bash profiles/device_run.sh <round-tag> <run-name> <stage> [<stage> ...]
# stages
#   suite          pytest -m gpu + __graft_entry__.smoke()
#   bench:N        the driver's command verbatim (`python3 bench.py --gpus 1 --steps 20 --warmup 5`) N times, every JSON line kept
#   benchlite:N    the same without cpu baseline / other configs (A/B of libraries: KORNIA_AMD_LIB is honoured)
#   prof           rocprofv3 --kernel-trace --stats of the bench command, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes
#   units          unit counters (separate --pmc passes) of the four hot launches + km_points -> <run-name>_units/ (pmc_units_to_json.py)
#   py:<script>    python <script> (a profiles/time_*.py), output appended to the run log
#   rocprof:<script>   the same under rocprofv3 --kernel-trace --stats, per-kernel table appended to the run log
#   ab:<script>:<lib1>,<lib2>,...   python <script> once per library under kornia_amd/lib/var (name without lib_/.so; "default" = the shipped one)
# Everything goes to gpurun_out/<round-tag>/<run-name>.txt (+ JSON lines in <run-name>_bench.jsonl, profiler output in <run-name>_prof/).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=$1; RUN=$2; shift 2
D=$R/gpurun_out/$TAG
mkdir -p $D
O=$D/$RUN.txt
: > $O
run() { echo "\$ $*" >> $O; timeout ${KM_STAGE_TIMEOUT:-900} "$@" >> $O 2>&1; local rc=$?; echo "[rc $rc]" >> $O; return $rc; }
for stage in "$@"; do
  echo "=== stage $stage ===" >> $O
  case $stage in
    suite)
      run python -m pytest tests -m gpu -x -q
      run python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
      ;;
    bench:*|benchlite:*)
      n=${stage#*:}
      extra=""; [ "${stage%%:*}" = benchlite ] && extra="--no-cpu-baseline --no-extras"
      for i in $(seq 1 $n); do
        BENCH_SETTLE_TRACE=1 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra > $D/.b.out 2> $D/.b.err
        echo "[bench $i rc $?]" >> $O
        grep "^settle" $D/.b.err | head -40 >> $O
        grep "^{" $D/.b.out | tail -1 >> $D/${RUN}_bench.jsonl
        grep "^{" $D/.b.out | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ms_per_step','ms_per_step_groups','longest_single_step_ms_per_group','ms_per_step_one_input_set','settle','clocks')}); print({k: v['ms'] for k, v in d['ops'].items()})" >> $O 2>&1
      done
      rm -f $D/.b.out $D/.b.err
      ;;
    prof)
      P=$D/${RUN}_prof; rm -rf $P; mkdir -p $P
      python3 -c "import bench; print(bench.csrc_sha256())" > $P/csrc.sha256 2>/dev/null
      ( cd /tmp && export TMPDIR=/tmp
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o f -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --groups 3 > $P/stats.log 2>&1
        rc=$?; echo "rocprofv3 stats rc $rc" >> $O
        if [ $rc -eq 0 ]; then
          for c in FETCH_SIZE WRITE_SIZE; do
            timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --groups 1 > $P/pmc_$c.log 2>&1
            rc=$?; echo "pmc $c rc $rc" >> $O
            [ $rc -ne 0 ] && break
          done
        fi )
      find $P -name "*_counter_collection.csv" | while read f; do head -1 $f > $f.km; grep "km_" $f >> $f.km; rm $f; done
      find $P -name "*_kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
      ;;
    py:*)
      run python ${stage#py:}
      ;;
    units)       # SQ / TA / TCP / TCC / GRBM counters of the hot kernels + km_points (profiles/pmc_units.sh, profiles/pmc_step.py)
      run bash profiles/pmc_units.sh $D/${RUN}_units
      ;;
    rocprof:*)   # rocprofv3 --kernel-trace --stats of python <script>; the km_* rows of the stats table go to the run log
      script=${stage#rocprof:}; P=$D/${RUN}_rocprof_$(basename $script .py); rm -rf $P; mkdir -p $P
      ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o f -- python $R/$script > $P/out.log 2>&1; echo "rocprofv3 rc $?" >> $O )
      grep "^lib=" $P/out.log >> $O
      find $P -name "*kernel_stats.csv" | head -1 | xargs -r python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f us' % (float(r['AverageNs']) / 1e3)).rjust(12), ('min %.1f' % (float(r['MinNs']) / 1e3)).rjust(12), ('max %.1f' % (float(r['MaxNs']) / 1e3)).rjust(12))
" >> $O 2>&1
      find $P -name "*_kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
      ;;
    ab:*)
      rest=${stage#ab:}; script=${rest%%:*}; libs=${rest#*:}
      for l in ${libs//,/ }; do
        if [ "$l" = default ]; then echo "--- library: default" >> $O; run python $script
        else echo "--- library: $l" >> $O; KORNIA_AMD_LIB=$R/kornia_amd/lib/var/lib_$l.so run python $script; fi
      done
      ;;
    *) echo "unknown stage $stage" >> $O ;;
  esac
done
grep -v "amdgpu.ids\|^\.\.\." $O | tail -${KM_TAIL:-60}
