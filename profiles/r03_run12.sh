#!/bin/bash
# round 3, device run 12: two workgroups of 512 threads per CU on 64 x 32 tiles (lib_th32), started in phase / out of phase (pseudo-random start
# delay up to 12 / 24 x 1024 cycles: lib_th32d, lib_th32d24) against the default (one workgroup of 1024 threads, 64 x 64 tiles); phase profiles
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run12.txt
: > $O
V=$PWD/kornia_amd/lib/var
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
for i in 1 2; do
  run python profiles/time_bwd_fused.py 20
  for v in th32 th32d th32d24; do KORNIA_AMD_LIB=$V/lib_$v.so run python profiles/time_bwd_fused.py 20; done
done
LAB_WORKERS=512 LAB_WAVES=8 LAB_TH=32 KORNIA_AMD_LIB=$V/lib_th32prof.so run python profiles/time_bwd_phases.py
LAB_WORKERS=512 LAB_WAVES=8 LAB_TH=32 KORNIA_AMD_LIB=$V/lib_th32dprof.so run python profiles/time_bwd_phases.py
grep -v "^{" $O | grep -v "amdgpu.ids" | tail -60
