#!/bin/bash
# LDS / wait counters of the backward kernels on the headline workload (two rocprofv3 --pmc
# passes, one run each), a phase-stamp run and a timing with the count + emit phases left out.
#   usage (on the GPU box): bash tools/r3_lds_pmc.sh [bench args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3_lds_pmc
mkdir -p "$out"
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/pmc_$name" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > /dev/null 2> "$out/pmc_$name.err"
  rc=$?
  echo "pass $name rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
BENCH_ARGS="$*"
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY
run_pass pipes SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS
python tools/pmc_table.py "$out" > "$out/summary.md" 2>&1
cat "$out/summary.md"
