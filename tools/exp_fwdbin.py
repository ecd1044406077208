"""What forward-time binning could buy: the binning pass of the backward (directory + scatter
records, no gradients) timed alone, beside the forward on a second stream, and the backward
without it.

    python tools/exp_fwdbin.py explibs/phase.so

Needs a build whose msda_backward takes the experiment bits 128 (gradients only: the workspace
already holds directory and records) and 256 (binning only)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness, _lib
w = harness.WORKLOADS[os.environ.get("WORKLOAD", "cityscapes_1024x2048_fp32_b8")]
value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", seed=1, dist=os.environ.get("DIST", "R"))
n, s, m, d = value.shape; lq, l, p = loc.shape[1], loc.shape[3], loc.shape[4]
code = {torch.float32: 0, torch.bfloat16: 2}
dv, dl = code[value.dtype], code[loc.dtype]
lib = ctypes.CDLL(os.path.abspath(sys.argv[1])); _lib._declare(lib)
host = (ctypes.c_int64 * (2 * l))(*[int(x) for x in shapes.flatten().tolist()])
vt = torch.empty((n, m, s, d), dtype=value.dtype, device="cuda")
out = torch.empty((n, lq, m * d), dtype=value.dtype, device="cuda")
gv = torch.empty_like(value); gl = torch.empty_like(loc); ga = torch.empty_like(attn)
HM = 8
nb = lib.msda_backward_workspace_bytes(n, s, m, d, lq, l, p, dv, dl, HM, host)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
lib.msda_value_relayout(value.data_ptr(), vt.data_ptr(), None, n, s, m, d, dv, 1, main.cuda_stream)
def fwd(st):
    rc = lib.msda_forward(vt.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), None, out.data_ptr(), n, s, m, d, lq, l, p, dv, dl, HM, st.cuda_stream); assert rc == 0, rc
def bwd(st, x):
    rc = lib.msda_backward(g.data_ptr(), vt.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), None, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), nb, n, s, m, d, lq, l, p, dv, dl, HM | x, host, st.cuda_stream); assert rc == 0, rc
def both():
    side.wait_stream(main)
    bwd(side, 256)
    fwd(main)
    main.wait_stream(side)
cases = {
    "forward": lambda: fwd(main),
    "binning alone": lambda: bwd(main, 256),
    "forward, then binning (one stream)": lambda: (fwd(main), bwd(main, 256)),
    "forward beside binning (two streams)": both,
    "backward": lambda: bwd(main, 0),
    "backward without binning": lambda: bwd(main, 128),
}
def timeit(f, n_it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n_it
bwd(main, 0); torch.cuda.synchronize()
for rnd in range(2):
    for name, f in cases.items():
        print(f"round {rnd} {name:40s} {timeit(f):.3f} ms", flush=True)
