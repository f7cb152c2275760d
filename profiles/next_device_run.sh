#!/bin/bash
# First device call of the next round: everything that was finished on the host build only, plus the timings that decide
# DESIGN.md section 7's list.  Writes under gpurun_out/ (copy what is worth keeping into profiles/).
#   gpurun --timeout 600 -- 'bash profiles/next_device_run.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
# 1. tests that have not been on a device yet (they sort last in the full run for the same reason)
timeout 240 python -m pytest tests/test_zz_gpu_registration.py tests/test_zz_gpu_canny.py tests/test_zz_gpu_fuzz_pyramid.py -q > gpurun_out/zz_tests.log 2>&1
echo "rc=$?" >> gpurun_out/zz_tests.log
# 2. fused loss (restructured kernel) and pyramid timings
timeout 60 python profiles/time_registration.py > gpurun_out/time_registration.txt 2>&1
timeout 60 python profiles/time_pyramid.py > gpurun_out/time_pyramid.txt 2>&1
# 3. the bench line with the informational block (pyrup / ScalePyramid / canny / config-5 one-launch step)
timeout 200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/zz_tests.log; cat gpurun_out/time_registration.txt gpurun_out/time_pyramid.txt; tail -c 1500 gpurun_out/bench.json
