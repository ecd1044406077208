"""A/B timing of several builds of libmsda.so in one process:

    python tools/exp_libs.py lib1.so lib2.so[:ENV=VAL] ...

ENV=VAL is set only around that variant's calls (the library reads its experiment switches from
the environment on every call).
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness, _lib
w = harness.WORKLOADS[os.environ.get("WORKLOAD", "cityscapes_1024x2048_fp32_b8")]
if os.environ.get("HALF"):  # the same pyramid at half the resolution, four times the batch
    import dataclasses
    w = dataclasses.replace(w, levels=tuple((h // 2, ww // 2) for h, ww in w.levels), batch=4 * w.batch)
DIST = os.environ.get("DIST", "R")
value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", seed=1, dist="R" if DIST == "ring" else DIST)
if DIST == "ring":
    # the layer's initial state: every query samples at whole-pixel offsets k * direction(head)
    # from its own pixel centre, uniform weights
    import math
    m_, l_, p_ = loc.shape[2], loc.shape[3], loc.shape[4]
    ang = torch.arange(m_, dtype=torch.float32, device="cuda") * (2 * math.pi / m_)
    ring = torch.stack((ang.cos(), ang.sin()), -1); ring = ring / ring.abs().amax(-1, keepdim=True)
    off = ring.view(m_, 1, 1, 2) * torch.arange(1, p_ + 1, device="cuda", dtype=torch.float32).view(1, 1, -1, 1)
    wh = torch.stack((shapes[:, 1], shapes[:, 0]), -1).float()
    ref = harness.encoder_reference_points(w, "cuda")
    loc = (ref.view(1, -1, 1, 1, 1, 2) + (off.view(1, 1, m_, 1, p_, 2) / wh.view(1, 1, 1, l_, 1, 2))).expand(loc.shape).to(loc.dtype).contiguous()
    attn = torch.full_like(attn, 1.0 / (l_ * p_))
def query_order(levels, spec):
    """QPERM=<th>x<tw>[c]: queries visited patch by patch (th x tw pixels of their own level,
    row-major inside), patches in raster order, or column by column with the 'c' suffix."""
    col = spec.endswith("c"); th, tw = (int(x) for x in spec.rstrip("c").split("x"))
    out, base = [], 0
    for h, w in levels:
        idx = torch.arange(h * w).view(h, w)
        ph, pw = -(-h // th), -(-w // tw)
        tiles = [(i, j) for j in range(pw) for i in range(ph)] if col else [(i, j) for i in range(ph) for j in range(pw)]
        for i, j in tiles:
            out.append(idx[i * th:(i + 1) * th, j * tw:(j + 1) * tw].reshape(-1) + base)
        base += h * w
    return torch.cat(out)
if os.environ.get("QPERM"):
    perm = query_order(w.levels, os.environ["QPERM"]).cuda()
    loc, attn, g = loc[:, perm].contiguous(), attn[:, perm].contiguous(), g[:, perm].contiguous()
    print("query order", os.environ["QPERM"], flush=True)
n, s, m, d = value.shape; lq, l, p = loc.shape[1], loc.shape[3], loc.shape[4]
code = {torch.float32: 0, torch.bfloat16: 2}
dv, dl = code[value.dtype], code[loc.dtype]
libs = {}
for spec in sys.argv[1:]:
    path = spec.partition(":")[0]
    lib = ctypes.CDLL(os.path.abspath(path)); _lib._declare(lib); libs[spec] = lib
host = (ctypes.c_int64 * (2 * l))(*[int(x) for x in shapes.flatten().tolist()])
vt = torch.empty((n, m, s, d), dtype=value.dtype, device="cuda")
out = torch.empty((n, lq, m * d), dtype=value.dtype, device="cuda")
gv = torch.empty_like(value); gl = torch.empty_like(loc); ga = torch.empty_like(attn)
HM = 0 if os.environ.get('NOHM') else 8
if not HM: vt = value
ws = None
def run(lib, what):
    global ws
    st = torch.cuda.current_stream().cuda_stream
    if what in ("relayout", "all") and HM:
        lib.msda_value_relayout(value.data_ptr(), vt.data_ptr(), None, n, s, m, d, dv, 1, st)
    if what in ("fwd", "all"):
        rc = lib.msda_forward(vt.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), None, out.data_ptr(), n, s, m, d, lq, l, p, dv, dl, HM, st); assert rc == 0, rc
    if what in ("bwd", "all"):
        nb = lib.msda_backward_workspace_bytes(n, s, m, d, lq, l, p, dv, dl, HM | int(os.environ.get('XFLAGS', '0')), host)
        if ws is None or ws.numel() < nb: ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        rc = lib.msda_backward(g.data_ptr(), vt.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), None, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), ws.numel(), n, s, m, d, lq, l, p, dv, dl, HM | int(os.environ.get('XFLAGS', '0')), host, st); assert rc == 0, rc
def set_env(spec, on):
    env = spec.partition(":")[2]
    if env:
        k, _, v = env.partition("=")
        if on: os.environ[k] = v
        else: os.environ.pop(k, None)
def timeit(lib, what, n_it=10):
    for _ in range(2): run(lib, what)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it): run(lib, what)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n_it
ref = None
for spec, lib in libs.items():
    set_env(spec, True); run(lib, "all"); set_env(spec, False)
    torch.cuda.synchronize()
    if ref is None: ref = (gv.clone(), gl.clone(), ga.clone(), out.clone())
    else: print(f"{spec}: max |diff| vs first variant: grad_value {(gv - ref[0]).abs().max().item():.3e} grad_loc {(gl - ref[1]).abs().max().item():.3e} grad_attn {(ga - ref[2]).abs().max().item():.3e} out {(out - ref[3]).abs().max().item():.3e}", flush=True)
for rnd in range(3):
    for spec, lib in libs.items():
        set_env(spec, True)
        print(f"round {rnd} {os.path.basename(spec):36s} fwd {timeit(lib,'fwd'):.3f}  bwd {timeit(lib,'bwd'):.3f}  all {timeit(lib,'all'):.3f} ms", flush=True)
        set_env(spec, False)
