#!/bin/bash
# build_variant_all.sh <tag> <arch> <extra flags...> : every source with another --offload-arch / extra flags -> kornia_amd/lib/var/lib_<tag>.so
set -e
tag=$1; arch=$2; shift; shift
cd /root/repo
mkdir -p kornia_amd/lib/var
F="--offload-arch=$arch -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
d=$(mktemp -d)
pids=""
for f in kornia_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  [ $b = km_warp_cubic ] && extra="-fno-slp-vectorize"
  hipcc $F $extra "$@" -c $f -o $d/$b.o & pids="$pids $!"
done
for p in $pids; do wait $p || { echo "compile FAILED"; rm -rf $d; exit 1; }; done
hipcc --offload-arch=$arch -shared -fPIC -o kornia_amd/lib/var/lib_$tag.so $d/*.o
rm -rf $d; echo built kornia_amd/lib/var/lib_$tag.so
