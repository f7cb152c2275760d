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
  hipcc $F "${FL[@]}" -c kornia_amd/csrc/$f.hip -o $d/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p || { echo "compile FAILED: no library written"; rm -rf $d; exit 1; }; done
objs=""
for o in kornia_amd/lib/obj/*.o; do b=$(basename $o); if [ -f $d/$b ]; then objs="$objs $d/$b"; else objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o kornia_amd/lib/var/lib_$tag.so $objs
rm -rf $d; echo built kornia_amd/lib/var/lib_$tag.so
