#!/bin/bash
# A/B of variant libraries (kornia_amd/lib/var/lib_<tag>.so) on the config-2 warp kernels: ab_libs.sh <which> <tag>...
which=$1; shift
mkdir -p gpurun_out
out=gpurun_out/ab_libs.txt
: > $out
echo "== default" >> $out
timeout 120 python profiles/time_warp_kernels.py 30 $which 2>&1 | grep -v amdgpu.ids >> $out
for tag in "$@"; do
  echo "== $tag" >> $out
  KORNIA_AMD_LIB=kornia_amd/lib/var/lib_$tag.so timeout 120 python profiles/time_warp_kernels.py 30 $which 2>&1 | grep -v amdgpu.ids >> $out
done
echo "== default again" >> $out
timeout 120 python profiles/time_warp_kernels.py 30 $which 2>&1 | grep -v amdgpu.ids >> $out
cat $out
