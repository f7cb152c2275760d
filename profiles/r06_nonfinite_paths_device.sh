#!/bin/bash
# every warp path with matrices that are not maps, in a process of its own (a device fault must not take a suite with it)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_zz_gpu_nonfinite_paths.py -m gpu -q -x > gpurun_out/r06/run8_nonfinite_paths.txt 2>&1
echo "[nonfinite paths rc $?]" >> gpurun_out/r06/run8_nonfinite_paths.txt
tail -15 gpurun_out/r06/run8_nonfinite_paths.txt
