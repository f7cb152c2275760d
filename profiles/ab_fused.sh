#!/bin/bash
# A/B of the backward of the warp at config 2: two launches (default) against the fused tile-owner form (KM_WARP_BWD_FUSE=1) in the
# variant libraries given as arguments (kornia_amd/lib/var/lib_<tag>.so, profiles/build_variant.sh).  gpurun -- 'bash profiles/ab_fused.sh gm1 gm2'
mkdir -p gpurun_out
out=gpurun_out/ab_fused.txt
: > $out
echo "== default library, two launches" >> $out
timeout 120 python profiles/time_warp_kernels.py 30 fwd,gm,sc,bwd >> $out 2>&1
for tag in "$@"; do
  echo "== $tag fused" >> $out
  KM_WARP_BWD_FUSE=1 KORNIA_AMD_LIB=kornia_amd/lib/var/lib_$tag.so timeout 120 python profiles/time_warp_kernels.py 30 bwd >> $out 2>&1
done
cat $out
