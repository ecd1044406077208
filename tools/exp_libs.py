"""A/B timing of several builds of libmsda.so in one process: python tools/exp_libs.py lib1.so lib2.so ..."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness, _lib
w = harness.WORKLOADS[os.environ.get("WORKLOAD", "cityscapes_1024x2048_fp32_b8")]
value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", seed=1)
n, s, m, d = value.shape; lq, l, p = loc.shape[1], loc.shape[3], loc.shape[4]
code = {torch.float32: 0, torch.bfloat16: 2}
dv, dl = code[value.dtype], code[loc.dtype]
libs = {}
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path)); _lib._declare(lib); libs[path] = lib
vt = torch.empty((n, m, s, d), dtype=value.dtype, device="cuda")
out = torch.empty((n, lq, m * d), dtype=value.dtype, device="cuda")
gv = torch.empty_like(value); gl = torch.empty_like(loc); ga = torch.empty_like(attn)
HM = 8
def run(lib, what):
    st = torch.cuda.current_stream().cuda_stream
    if what in ("relayout", "all"):
        lib.msda_value_relayout(value.data_ptr(), vt.data_ptr(), n, s, m, d, dv, 1, st)
    if what in ("fwd", "all"):
        rc = lib.msda_forward(vt.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), out.data_ptr(), n, s, m, d, lq, l, p, dv, dl, HM, st); assert rc == 0, rc
    if what in ("bwd", "all"):
        nb = lib.msda_backward_workspace_bytes(n, s, m, d, lq, l, p, dv, dl, HM)
        global ws
        if ws is None or ws.numel() < nb: ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        rc = lib.msda_backward(g.data_ptr(), vt.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), ws.numel(), n, s, m, d, lq, l, p, dv, dl, HM, st); assert rc == 0, rc
ws = None
def timeit(lib, what, n_it=10):
    for _ in range(2): run(lib, what)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it): run(lib, what)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n_it
for lib in libs.values(): run(lib, "all")
for rnd in range(3):
    for path, lib in libs.items():
        print(f"round {rnd} {os.path.basename(path):24s} fwd {timeit(lib,'fwd'):.3f}  bwd {timeit(lib,'bwd'):.3f}  all {timeit(lib,'all'):.3f} ms", flush=True)
