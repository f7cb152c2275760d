#!/bin/bash
# build_variant.sh <tag> <flags...> : rebuilds the three warp sources with extra -D flags, links with the current objects -> scratch/r2/lib_<tag>.so
set -e
tag=$1; shift
cd /root/repo
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
d=$(mktemp -d)
for f in km_warp km_warp_gm km_warp_bwd_tiled; do hipcc $F "$@" -c kornia_amd/csrc/$f.hip -o $d/$f.o & done
wait
objs=$(ls kornia_amd/lib/obj/*.o | grep -v "/km_warp.o\|/km_warp_gm.o\|/km_warp_bwd_tiled.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/r2/lib_$tag.so $objs $d/km_warp.o $d/km_warp_gm.o $d/km_warp_bwd_tiled.o
rm -rf $d; echo built scratch/r2/lib_$tag.so
