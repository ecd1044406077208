"""Does the tile kernel of batch part k hide under the query kernel of part k + 1?

    python tools/exp_overlap.py explibs/phases.so        (a -DMSDA_EXP_PHASES=1 build)

The binned backward is two kernels: the query kernel (bound by the vector-memory path) and the
tile kernel (bound by vector-ALU issue).  The batch is cut into K parts; part k's tile kernel is
issued on a second stream behind an event, next to part k + 1's query kernel on the first.
Prints ms per backward for K = 1 (the plain call), and for K = 2, 4, 8 serial / overlapped.
"""
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
HM = 8
NO_TILE, NO_QUERY = 1 << 20, 1 << 21
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
lib.msda_value_relayout(value.data_ptr(), vt.data_ptr(), None, n, s, m, d, dv, 1, main.cuda_stream)
gv = torch.empty_like(value); gl = torch.empty_like(loc); ga = torch.empty_like(attn)


def call(b0, nb, ws, flags, stream):
    rc = lib.msda_backward(g[b0:].data_ptr(), vt[b0:].data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
                           loc[b0:].data_ptr(), attn[b0:].data_ptr(), None, gv[b0:].data_ptr(),
                           gl[b0:].data_ptr(), ga[b0:].data_ptr(), ws.data_ptr(), ws.numel(), nb, s, m, d,
                           lq, l, p, dv, dl, HM | flags, host, stream.cuda_stream)
    assert rc == 0, rc


def make_ws(nb, parts):
    need = lib.msda_backward_workspace_bytes(nb, s, m, d, lq, l, p, dv, dl, HM, host)
    return [torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(parts)]


def plain(ws):
    call(0, n, ws[0], 0, main)


def serial(k, ws):
    nb = n // k
    for i in range(k):
        call(i * nb, nb, ws[i], NO_TILE, main)
        call(i * nb, nb, ws[i], NO_QUERY, main)


def overlapped(k, ws, evs):
    nb = n // k
    for i in range(k):
        call(i * nb, nb, ws[i], NO_TILE, main)
        evs[i].record(main)
        side.wait_event(evs[i])
        call(i * nb, nb, ws[i], NO_QUERY, side)
    done = evs[k]
    done.record(side)
    main.wait_event(done)


def timeit(fn, n_it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_it


ws1 = make_ws(n, 1)
plain(ws1); torch.cuda.synchronize()
ref = (gv.clone(), gl.clone(), ga.clone())
for rnd in range(2):
    print(f"round {rnd}: plain {timeit(lambda: plain(ws1)):.3f} ms", flush=True)
    for k in (1, 2, 4, 8):
        if n % k: continue
        ws = make_ws(n // k, k)
        evs = [torch.cuda.Event() for _ in range(k + 1)]
        gv.zero_(); gl.zero_(); ga.zero_()
        overlapped(k, ws, evs); torch.cuda.synchronize()
        diff = max((gv - ref[0]).abs().max().item(), (gl - ref[1]).abs().max().item(), (ga - ref[2]).abs().max().item())
        ts = timeit(lambda: serial(k, ws)); to = timeit(lambda: overlapped(k, ws, evs))
        print(f"  K={k}: serial {ts:.3f}  overlapped {to:.3f} ms   max |diff| vs plain {diff:.2e}", flush=True)
