#!/bin/bash
# A/B of an environment switch on the whole config-2 step: ab_env.sh VAR=value [VAR=value ...]  (each is compared with the default, interleaved)
mkdir -p gpurun_out
out=gpurun_out/ab_env.txt
: > $out
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  echo "== default" >> $out; run >> $out
  for kv in "$@"; do echo "== $kv" >> $out; env $kv bash -c "$(declare -f run); run" >> $out; done
done
cat $out
