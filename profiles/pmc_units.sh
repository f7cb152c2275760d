#!/bin/bash
# SQ / TA / TCP / GRBM unit counters of the hot kernels (and of km_points as the control), separate --pmc passes with --kernel-trace only
# (each under its own timeout), on profiles/pmc_step.py:   bash profiles/pmc_units.sh <outdir>     then profiles/pmc_units_to_json.py <tag> <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$n -o p -- python $R/profiles/pmc_step.py 2 > $OUT/pmc_$n.log 2>&1; echo "pmc $n rc $?"; }
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass sq3 SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT
pass ta TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum
pass tcp TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass tcc2 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
find $OUT -name "*_counter_collection.csv" | while read f; do head -1 $f > $f.km; grep "km_" $f >> $f.km; rm $f; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT
