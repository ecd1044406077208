import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness
from unipercept_amd.deform_attn import _ops
w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", seed=1)
fwd = torch.ops.unipercept.ms_deform_attn_fwd
def timeit(v, n=20):
    for _ in range(5): fwd(v, shapes, lsi, loc, attn, 0)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fwd(v, shapes, lsi, loc, attn, 0)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
ref = fwd(value, shapes, lsi, loc, attn, 0)
print("token-major ms", timeit(value))
os.environ["MSDA_EXP_HEADMAJOR"]="1"
vt = value.permute(0,2,1,3).contiguous()           # (N,M,S,D) in memory
vt_as = vt.view(value.shape)                         # same bytes, API shape
out = fwd(vt_as, shapes, lsi, loc, attn, 0)
print("head-major ms", timeit(vt_as), "maxdiff", float((out-ref).abs().max()))
t0=torch.cuda.Event(enable_timing=True); t1=torch.cuda.Event(enable_timing=True)
t0.record(); 
for _ in range(10): vt = value.permute(0,2,1,3).contiguous()
t1.record(); torch.cuda.synchronize(); print("torch transpose ms", t0.elapsed_time(t1)/10)
