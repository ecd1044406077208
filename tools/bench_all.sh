set -e
run() { name=$1; shift; timeout -k 10 150 python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/r2_$name.json 2> gpurun_out/r2_$name.err; python - "$name" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r2_{sys.argv[1]}.json")); p=d["phases"]
print(sys.argv[1], round(p["fwd_ms"],3), round(p["bwd_ms"],3), round(d["ms_per_step"],3), round(d["value"]/1e9,2), round(p["step_hbm_frac"]*100,2), flush=True)
PY
}
run fp32_R
run fp32_U --dist U
run bf16 --workload cityscapes_1024x2048_bf16_b8
run bf16_locf32 --workload cityscapes_1024x2048_bf16_locf32_b8
run coco --workload coco_1024x1024_L4_bf16_b16
run dvps_fp32 --workload dvps_1024x2048_L4_P8_fp32_b8
run dvps_bf16 --workload dvps_1024x2048_L4_P8_bf16_b8
run packed --packed
run layer --layer
