#!/bin/bash
# round 3, device run 14: the one-read backward with fewer persistent workgroups than CUs (256 / 192 / 128 / 64): cycles per tile and the clock
# they run at - is the kernel traded against the chip's power budget?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run14.txt
: > $O
V=$PWD/kornia_amd/lib/var
for w in 256 192 128 64 256; do
  echo "== $w workers" >> $O
  LAB_WORKERS=$w KORNIA_AMD_LIB=$V/lib_profw$w.so timeout 300 python profiles/time_bwd_phases.py 2>&1 | grep -v amdgpu >> $O
done
cat $O
