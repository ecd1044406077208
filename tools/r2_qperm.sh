#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for qp in "" 8x8 4x16 16x16 12x16 8x24 128x16c 128x32c 128x64c 16x16c; do
  echo "== QPERM=$qp" | tee -a gpurun_out/qperm.log
  QPERM=$qp timeout -k 10 200 python -u tools/exp_libs.py "$@" 2>&1 | grep -E "round 2" | tee -a gpurun_out/qperm.log || exit 1
done
