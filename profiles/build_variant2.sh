#!/bin/bash
# build_variant2.sh <tag> <file>:<flags,comma-separated> ... : per-file -D flags, links with the current objects -> kornia_amd/lib/var/lib_<tag>.so
set -e
tag=$1; shift
cd /root/repo
mkdir -p kornia_amd/lib/var
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
d=$(mktemp -d)
for spec in "$@"; do
  f=${spec%%:*}; fl=${spec#*:}
  IFS='|' read -ra FL <<< "$fl"
  # the per-file flags of the shipped build (kornia_amd/build.py FILE_FLAGS: -fno-slp-vectorize for km_warp_bwd_fused / km_warp_cubic) come first:
  # runs 28-36 of round 4 were built WITHOUT them (the one-read backward is 8 % slower with the SLP vectorizer's packed pairs) - run 37 repeated the kept ones with them
  PF=$(python3 -c "import sys; sys.path.insert(0, 'kornia_amd'); import build; print(' '.join(build.FILE_FLAGS.get('$f.hip', [])))" 2>/dev/null)
  hipcc $F $PF "${FL[@]}" -c kornia_amd/csrc/$f.hip -o $d/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p || { echo "compile FAILED: no library written"; rm -rf $d; exit 1; }; done
objs=""
for o in kornia_amd/lib/obj/*.o; do b=$(basename $o); if [ -f $d/$b ]; then objs="$objs $d/$b"; else objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o kornia_amd/lib/var/lib_$tag.so $objs
rm -rf $d; echo built kornia_amd/lib/var/lib_$tag.so
