"""Timing: weight gradient of a linear layer over K = 344064 tokens, plain GEMM vs split-K bmm."""
import torch
k, c, o = 8 * 43008, 256, 256
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(k, c, device="cuda", dtype=dt)
    for o in (256, 288):
        g = torch.randn(k, o, device="cuda", dtype=dt)
        def t(fn, it=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
        plain = lambda: g.t() @ x
        res = [f"plain {t(plain):.3f}"]
        ref = plain()
        for sk in (8, 16, 32, 64, 128):
            if k % sk: continue
            fn = lambda: torch.bmm(g.view(sk, k // sk, o).transpose(1, 2), x.view(sk, k // sk, c)).sum(0)
            err = (fn().float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
            res.append(f"sk{sk} {t(fn):.3f} (rel {err:.1e})")
        bias = lambda: g.sum(0)
        ones = torch.ones(k, device="cuda", dtype=dt)
        bias2 = lambda: ones @ g
        res.append(f"bias sum {t(bias):.3f}  ones@g {t(bias2):.3f}")
        print(dt, o, " | ".join(res), flush=True)
