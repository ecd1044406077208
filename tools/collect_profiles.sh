#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box.
#   usage: tools/collect_profiles.sh <tag> [bench args...]
# Writes raw CSVs under gpurun_out/prof_<tag>/ and a summary JSON/markdown next to them;
# copy the summaries into profiles/ afterwards.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
mkdir -p "$out"
args="--steps 10 --warmup 3 --no-cpu-baseline $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python bench.py $args > "$out/bench_under_trace.json" 2> "$out/kt.err" || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
    "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$out/pmc_$name.err" || exit 1
done
timeout -k 10 200 python bench.py --steps 50 --warmup 10 "$@" > "$out/bench.json" 2> "$out/bench.err" || exit 1
python tools/summarize_profiles.py "$out" "$tag"
