#!/bin/bash
# round 3, device run 15: the forward's last 25 / 40 / 100 % of blocks write with plain instead of streaming stores (so that the blur, which walks the
# batch the other way round, may find what was written last in the Infinity Cache): the step and the blur's time in it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run15.txt
: > $O
V=$PWD/kornia_amd/lib/var
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
for i in 1 2; do
  run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
  for v in tail25 tail40 tail100; do KORNIA_AMD_LIB=$V/lib_$v.so run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras; done
done
grep "^{" $O | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['ops'].items()})"
