#!/bin/bash
# round 6, run 3: the scan beside the persistent loop (forked stream) against the same kernel on the launch stream, then the suite and the bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; D=$R/gpurun_out/r06; mkdir -p $D
O=$D/run3_scan_ab.txt; : > $O
for i in 1 2; do
  for l in default scan_serial; do
    echo "--- library: $l" >> $O
    if [ $l = default ]; then unset KORNIA_AMD_LIB; else export KORNIA_AMD_LIB=$R/kornia_amd/lib/var/lib_$l.so; fi
    timeout 300 python profiles/time_bwd_fused.py >> $O 2>&1
    timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['ms_per_step_groups'], {k: v['ms'] for k, v in d['ops'].items()})" >> $O 2>&1
  done
done
unset KORNIA_AMD_LIB
tail -20 $O
KM_STAGE_TIMEOUT=1500 bash profiles/device_run.sh r06 run3 suite bench:2
