#!/bin/bash
# unit counters of the one-read backward (and the two launches it replaces), separate rocprofv3 --pmc passes (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03/pmc_fused
mkdir -p $O
B="python $R/profiles/time_bwd_fused.py 3"
pass() { n=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $B > $O/$n.log 2>&1; echo "$n rc $?"; }
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
pass ta TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum
pass tcp TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
pass grbm GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# keep only the rows of the backward kernels (the CSVs hold every launch of the process)
for d in sq sq2 sq3 ta tcp grbm fetch write; do
  f=$O/$d/p_counter_collection.csv
  [ -f $f ] && { head -1 $f > $O/$d.csv; grep "km_warp_bwd\|km_warp_gm" $f >> $O/$d.csv; rm -rf $O/$d; }
done
ls -la $O | head -30
