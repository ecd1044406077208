import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness, _lib
from unipercept_amd.deform_attn import _ops
w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", seed=1)
vt = torch.ops.unipercept.ms_deform_attn_value_head_major(value)
bwd = torch.ops.unipercept.ms_deform_attn_bwd
HM = _lib.FLAG_VALUE_HEAD_MAJOR
def timeit(n=10):
    for _ in range(3): bwd(g, vt, shapes, lsi, loc, attn, HM)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): bwd(g, vt, shapes, lsi, loc, attn, HM)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for ab in [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 7, 8, 8|32, 8|64, 8|32|64]:
    os.environ["MSDA_EXP_ABLATE"]=str(ab)
    print(f"ablate={ab:3d}: bwd total {timeit():.3f} ms", flush=True)
