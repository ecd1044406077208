"""Timing: value projection written head-major (N,M,S,D) by batched GEMMs vs nn.Linear + re-layout."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd.deform_attn import _ops  # noqa
n, s, c, m = 8, 43008, 256, 8
d = c // m
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(n, s, c, device="cuda", dtype=dt)
    w = torch.randn(c, c, device="cuda", dtype=dt) * 0.05
    b = torch.randn(c, device="cuda", dtype=dt)
    def t(fn, it=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
    lin = lambda: torch.nn.functional.linear(x, w, b)
    def lin_relayout():
        v = torch.nn.functional.linear(x, w, b).view(n, s, m, d)
        return torch.ops.unipercept.ms_deform_attn_value_head_major(v)
    wt = w.view(m, d, c).transpose(1, 2)  # (M, C, D)
    bb = b.view(m, 1, d)
    out = torch.empty(n, m, s, d, device="cuda", dtype=dt)
    def bmm_loop():
        for i in range(n):
            torch.baddbmm(bb, x[i].expand(m, s, c), wt, out=out[i])
        return out
    def bmm_one():
        # (N*... ) one call: x viewed (1, N*S, C) expanded over heads -> (M, N*S, D), i.e. (M, N, S, D)
        return torch.baddbmm(bb, x.view(1, n * s, c).expand(m, n * s, c), wt)
    ref = lin_relayout(); got = bmm_loop()
    print(dt, "max diff", (ref.float() - got.float()).abs().max().item())
    print(dt, "linear %.3f  linear+relayout %.3f  baddbmm loop %.3f  baddbmm one (M,N,S,D) %.3f ms" % (t(lin), t(lin_relayout), t(bmm_loop), t(bmm_one)))
