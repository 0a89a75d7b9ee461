#!/bin/bash
# rocprofv3 --pmc passes over tools/gather_replay.py (A/B build), one counter group per pass, for the kernels named in
# GR_ONLY (default: old,run).  Summaries land in gpurun_out/gather_pmc.txt.   usage: bash tools/gather_pmc.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NGP_AB_VARIANTS=1
export GR_ONLY=${GR_ONLY:-old,run}
OUT=gpurun_out/gather_pmc.txt
: > $OUT
i=0
for grp in \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_EA0_RDREQ_32B_sum" ; do
  i=$((i+1))
  rm -rf /tmp/gp_$i
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-include-regex 'grid_(fwd|bwd_input)' --output-format csv -d /tmp/gp_$i -- python3 tools/gather_replay.py > gpurun_out/gather_pmc_$i.log 2>&1
  echo "## pass $i: $grp (rc $?)" >> $OUT
  python3 tools/pmc_summary.py "/tmp/gp_$i/**/*counter_collection.csv" grid_ >> $OUT
done
tail -4 gpurun_out/gather_pmc_1.log >> $OUT
cat $OUT
