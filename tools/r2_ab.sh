#!/bin/bash
# A/B of library builds in one process per workload: tools/r2_ab.sh <tag> lib1 lib2 ...
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for wl in ${WORKLOADS:-cityscapes_1024x2048_fp32_b8 cityscapes_1024x2048_bf16_b8 coco_1024x1024_L4_bf16_b16}; do
  echo "== $wl" | tee -a gpurun_out/ab_$tag.log
  WORKLOAD=$wl timeout -k 10 240 python -u tools/exp_libs.py "$@" >> gpurun_out/ab_$tag.log 2>&1 || { tail -5 gpurun_out/ab_$tag.log; exit 1; }
done
grep -E "^==|round 2|diff" gpurun_out/ab_$tag.log
