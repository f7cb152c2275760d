#!/bin/bash
# builds the library of a git revision into kornia_amd/lib/var/lib_<tag>.so (A/B timing on one box): build_revision.sh <rev> <tag>
set -e
rev=$1; tag=$2
cd /root/repo
tmp=$(mktemp -d)
git archive $rev kornia_amd/csrc | tar -x -C $tmp
objs=""
for f in $tmp/kornia_amd/csrc/*.hip; do
  o=$tmp/$(basename $f .hip).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -c $f -o $o &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o kornia_amd/lib/var/lib_${tag}.so $objs
rm -rf $tmp
echo built kornia_amd/lib/var/lib_${tag}.so
