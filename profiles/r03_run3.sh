#!/bin/bash
# round 3, device run 3: new device tests, same-box A/B of the step with the one-read backward on / off, guarded rocprofv3 passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03
O=$R/gpurun_out/r03/run3.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python -m pytest tests/test_gpu_warp_fused.py tests/test_gpu_edge_cases.py tests/test_gpu_filters.py tests/test_zz_gpu_fuzz_pyramid.py -m gpu -x -q
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
for i in 1 2; do
  KM_WARP_BWD_FUSED=0 run $B
  run $B
done
run python profiles/time_transform_points.py
# ---- rocprofv3, guarded: the first pass decides whether the others run ----
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/r03/prof
mkdir -p $P
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o f -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $P/stats.log 2>&1
rc=$?; echo "rocprofv3 stats rc $rc" >> $O
if [ $rc -eq 0 ]; then
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/tp_stats -o f -- python $R/profiles/time_transform_points.py 6 > $P/tp_stats.log 2>&1; echo "tp stats rc $?" >> $O
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $P/pmc_$c.log 2>&1; rc=$?; echo "pmc $c rc $rc" >> $O
    [ $rc -ne 0 ] && break
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/tp_pmc_$c -o p -- python $R/profiles/time_transform_points.py 3 > $P/tp_pmc_$c.log 2>&1; echo "tp pmc $c rc $?" >> $O
  done
fi
# keep the merged output small: stats CSVs and the km_ rows of the counter CSVs
find $P -name "*_counter_collection.csv" | while read f; do head -1 $f > $f.km; grep "km_" $f >> $f.km; rm $f; done
find $P -name "*_kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
cd $R
# ---- two ranks on the one GPU of this box over RCCL (the multi-GPU legs on HIP tensors; no scaling claim) ----
export BENCH_FORCE_DEVICE=0 NCCL_DEBUG=WARN
for mode in all_gather p2p chunked; do
  echo "\$ 2 ranks, gather=$mode" >> $O
  timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --batch 32 --steps 5 --warmup 2 --scaling strong --gather $mode --no-cpu-baseline --no-extras >> $O 2>&1
  echo "[rc $?]" >> $O
done
grep -v "^{" $O | tail -40
