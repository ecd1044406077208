"""Function-level timing of MultiScaleFlashAttnFunction: kernels reading the packed tensor in place
vs the unpack / re-pack route (MSDA_NO_PACKED_POINTS=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness
from unipercept_amd.deform_attn import MultiScaleFlashAttnFunction, MultiScaleDeformAttnFunction
w = harness.WORKLOADS[os.environ.get("WORKLOAD", harness.HEADLINE)]
value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", seed=1)
packed = torch.cat([loc.flatten(-3), attn.flatten(-2)], -1).contiguous()
def step_flash():
    v = value.detach().requires_grad_(True); pk = packed.detach().requires_grad_(True)
    MultiScaleFlashAttnFunction.apply(v, shapes, lsi, pk, 64).backward(g)
def step_plain():
    v = value.detach().requires_grad_(True); lo = loc.detach().requires_grad_(True); a = attn.detach().requires_grad_(True)
    MultiScaleDeformAttnFunction.apply(v, shapes, lsi, lo, a, 64).backward(g)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
print(f"MultiScaleDeformAttnFunction fwd+bwd (autograd): {timeit(step_plain):.3f} ms")
print(f"MultiScaleFlashAttnFunction, packed read in place: {timeit(step_flash):.3f} ms")
os.environ["MSDA_NO_PACKED_POINTS"] = "1"
print(f"MultiScaleFlashAttnFunction, unpack + re-pack route: {timeit(step_flash):.3f} ms")
