#!/bin/bash
# A/B of variant libraries on the whole config-2 step (bench.py without the CPU baseline and the extras): ab_step.sh <tag>...
mkdir -p gpurun_out
out=gpurun_out/ab_step.txt
: > $out
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
echo "== default" >> $out; run >> $out
for tag in "$@"; do echo "== $tag" >> $out; KORNIA_AMD_LIB=kornia_amd/lib/var/lib_$tag.so run >> $out; done
echo "== default again" >> $out; run >> $out
cat $out
