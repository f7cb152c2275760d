#!/bin/bash
# round 3, device run 4: the whole device suite, the backward under rotations (general launch), config-3 timings, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03
O=$R/gpurun_out/r03/run4.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 900 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python -m pytest tests -m gpu -x -q
run python profiles/time_bwd_rotated.py 10
run python profiles/time_config3.py
run python bench.py --steps 20 --warmup 5
grep -v "^{" $O | tail -40
