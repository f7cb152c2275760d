#!/bin/bash
# round 3, device run 11: the register-tiled blur with vertically adjacent strips walking in opposite directions (shared halo rows touched at the
# same moment) against every strip walking down (lib_bluralt0): step, per-op times, HBM traffic of both
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03
O=$R/gpurun_out/r03/run11.txt
: > $O
V=$R/kornia_amd/lib/var
run() { echo "\$ $*" >> $O; timeout 600 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python -m pytest tests/test_gpu_filters.py tests/test_gpu_golden.py tests/test_gpu_config_parity.py tests/test_gpu_augmentation.py tests/test_gpu_edge_cases.py -m gpu -x -q
for i in 1 2; do
  run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
  KORNIA_AMD_LIB=$V/lib_bluralt0.so run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
done
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/r03/prof11
rm -rf $P; mkdir -p $P
for lib in default bluralt0; do
  [ $lib = default ] && unset KORNIA_AMD_LIB || export KORNIA_AMD_LIB=$V/lib_$lib.so
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch_$lib -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $P/fetch_$lib.log 2>&1; echo "pmc FETCH $lib rc $?" >> $O
done
unset KORNIA_AMD_LIB
find $P -name "*_counter_collection.csv" | while read f; do head -1 $f > $f.km; grep "km_blur" $f >> $f.km; rm $f; done
find $P -name "*_kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
cd $R
for lib in default bluralt0; do python - <<PY >> $O
import csv
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("$P/fetch_$lib/p_counter_collection.csv.km")) if "5, false" in r["Kernel_Name"]]
w=[float(r["Counter_Value"]) for r in csv.DictReader(open("$P/fetch_$lib/p_counter_collection.csv.km")) if "5, true" in r["Kernel_Name"]]
print("$lib: blur fwd FETCH_SIZE x2 = %.4f GB (%d launches), adjoint %.4f GB" % (2*1024*sum(v)/len(v)/1e9, len(v), 2*1024*sum(w)/len(w)/1e9))
PY
done
grep -v "^{" $O | grep -v "amdgpu.ids\|^\.\.\." | tail -20
grep "^{" $O | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['ops'].items()})"
