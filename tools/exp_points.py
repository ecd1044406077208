"""A/B timing of the sampling-point epilogue kernels of several libmsda.so builds."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness, _lib
w = harness.WORKLOADS[os.environ.get("WORKLOAD", harness.HEADLINE)]
dt = harness.torch_dtype(w.dtype); code = {torch.float32: 0, torch.bfloat16: 2}[dt]
shapes, _ = harness.level_tensors(w, "cuda")
n, s, m, l, p = w.batch, w.positions, w.heads, w.n_levels, w.points
raw = torch.randn(n, s, m, 3 * l * p, device="cuda").to(dt); ref = torch.rand(n, s, l, 2, device="cuda").to(dt)
out = torch.empty_like(raw); g = torch.randn_like(raw); graw = torch.empty_like(raw)
nbytes = raw.numel() * raw.element_size()
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path)); _lib._declare(lib)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.msda_points_forward(raw.data_ptr(), ref.data_ptr(), shapes.data_ptr(), out.data_ptr(), n, s, m, l, p, code, st)
    b = lambda: lib.msda_points_backward(g.data_ptr(), out.data_ptr(), shapes.data_ptr(), graw.data_ptr(), n, s, m, l, p, code, st)
    assert f() == 0 and b() == 0
    tf, tb = timeit(f), timeit(b)
    print(f"{os.path.basename(path):16s} fwd {tf:.3f} ms ({2*nbytes/tf/1e6:.0f} GB/s)  bwd {tb:.3f} ms ({3*nbytes/tb/1e6:.0f} GB/s)", flush=True)
