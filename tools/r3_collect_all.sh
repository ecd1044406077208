#!/bin/bash
# Evidence (round 3 kernels) for every benchmark configuration: one tools/collect_profiles.sh run per tag (set $1 = a or b for the two halves).
cd "$GRAFT_REPO_ROOT" || exit 1
if [ "$1" = a ]; then
  bash tools/collect_profiles.sh r03_cityscapes_fp32_b8 &&
  bash tools/collect_profiles.sh r03_cityscapes_fp32_b8_U --dist U &&
  bash tools/collect_profiles.sh r03_cityscapes_bf16_b8 --workload cityscapes_1024x2048_bf16_b8
else
  bash tools/collect_profiles.sh r03_coco_L4_bf16_b16 --workload coco_1024x1024_L4_bf16_b16 &&
  bash tools/collect_profiles.sh r03_dvps_L4_P8_fp32_b8 --workload dvps_1024x2048_L4_P8_fp32_b8 &&
  bash tools/collect_profiles.sh r03_dvps_L4_P8_bf16_b8 --workload dvps_1024x2048_L4_P8_bf16_b8
fi
