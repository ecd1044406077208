import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness
from unipercept_amd.deform_attn import _ops
w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", seed=1)
bwd = torch.ops.unipercept.ms_deform_attn_bwd
def timeit(n=10):
    for _ in range(3): bwd(g, value, shapes, lsi, loc, attn, 0)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): bwd(g, value, shapes, lsi, loc, attn, 0)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for ab in [0, 1, 2, 4, 1|2, 1|2|4, 8, 16, 8|16, 1|8|16]:
    os.environ["MSDA_EXP_ABLATE"]=str(ab)
    print(f"ablate={ab:2d}: bwd total {timeit():.3f} ms", flush=True)
