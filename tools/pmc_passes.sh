#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> -- one rocprofv3 --pmc run per counter group over a short bench
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
i=0
while read -r pass; do
  [ -z "$pass" ] && continue
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d "$out/p$i" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$out/p$i.err" || { echo "pass $i failed"; exit 1; }
  echo "== $pass"; python tools/pmc_summary.py "$out/p$i"
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
PASSES
