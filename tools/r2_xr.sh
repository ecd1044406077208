#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for wl in cityscapes_1024x2048_fp32_b8 cityscapes_1024x2048_bf16_b8; do
  for nohm in "" 1; do
    echo "== $wl NOHM=$nohm" | tee -a gpurun_out/xr.log
    NOHM=$nohm WORKLOAD=$wl timeout -k 10 200 python -u tools/exp_libs.py explibs/cur.so explibs/xr.so 2>&1 | grep -E "round 2|diff" | tee -a gpurun_out/xr.log
  done
done
