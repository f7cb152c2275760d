#!/bin/bash
# A/B of a variant library on the step AND the other configs (bench.py with its extras, without the CPU baseline): ab_extras.sh <tag>...
mkdir -p gpurun_out
out=gpurun_out/ab_extras.txt
: > $out
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
oc=d['other_configs']
for k,v in oc.items(): print('   ', k, v if not isinstance(v,dict) else {a:b for a,b in v.items() if a!='mfma'})"; }
echo "== default" >> $out; run >> $out
for tag in "$@"; do echo "== $tag" >> $out; KORNIA_AMD_LIB=kornia_amd/lib/var/lib_$tag.so run >> $out; done
cat $out
