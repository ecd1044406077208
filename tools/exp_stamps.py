"""Phase breakdown of the two backward kernels from a -DMSDA_STAMPS=1 build of the library:
    python tools/exp_stamps.py explibs/stamps.so
Prints, per kernel, the mean cycles thread 0 of a workgroup spends in every phase."""
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
gv = torch.empty_like(value); gl = torch.empty_like(loc); ga = torch.empty_like(attn)
st = torch.cuda.current_stream().cuda_stream
lib.msda_value_relayout(value.data_ptr(), vt.data_ptr(), None, n, s, m, d, dv, 1, st)
nb = lib.msda_backward_workspace_bytes(n, s, m, d, lq, l, p, dv, dl, 8, host)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")  # ablation builds leave records unwritten: zeros are valid records
for _ in range(3):
    rc = lib.msda_backward(g.data_ptr(), vt.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), None, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), nb, n, s, m, d, lq, l, p, dv, dl, 8, host, st); assert rc == 0
torch.cuda.synchronize()
import numpy as np
G = 1 << 15
buf = (ctypes.c_ulonglong * (G * 16))()
names = [["levels+tilemap", "issue point loads", "zero counters", "count (waits loads)", "scan+dir row", "park + stage first chunk (after emit)", "barrier after stage", "go row load+copy", "gather+dots+stores", "barrier after gather", "emit: records into the LDS image", "emit: barrier", "emit: linear copy + barrier"],
         ["setup: dir column + run scan", "A records+count own (fused: first rounds of all patches)", "barrier A", "B scan", "C fill own", "barrier C", "D walk own", "corner exchange + store"]]
for which, kname in ((0, "bwd_query_kernel"), (1, "bwd_tile_gather_kernel")):
    lib.msda_debug_read_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    got = lib.msda_debug_read_stamps(which, buf, G)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(G, 16).astype(np.float64)
    if which == 1 and os.environ.get("TILE") and not os.environ.get("NOLEVELS"):
        # per level: workgroup (batch, tile*M + head), coarse levels first
        t = int(os.environ["TILE"]); per = [(-(-h // t)) * (-(-ww // t)) for h, ww in w.levels]
        gx = sum(per) * m; life = a.sum(1)
        for lvl in range(len(per)):
            hi = sum(per[lvl:]) ; lo = sum(per[lvl + 1:])   # reversed tile order
            idx = [by * gx + bx for by in range(n) for bx in range(lo * m, hi * m) if by * gx + bx < G]
            sel = life[idx]; sel = sel[sel > 0]
            if len(sel): print(f"   level {lvl}: {len(sel)} workgroups, lifetime mean {sel.mean():.0f} max {sel.max():.0f}")
    a = a[a.sum(1) > 0]
    tot = a.sum(1)
    print(f"{kname}: {len(a)} workgroups sampled, mean lifetime {tot.mean():.0f} cycles (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
    for i, nm in enumerate(names[which]):
        print(f"   {nm:32s} {a[:, i].mean():9.0f}  {100 * a[:, i].mean() / tot.mean():5.1f} %")
