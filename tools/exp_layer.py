"""MSDeformAttn layer fwd+bwd at the headline shapes: fused point path vs separate projections."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd import harness
from unipercept_amd.deform_attn import MSDeformAttn
w = harness.WORKLOADS[os.environ.get("WORKLOAD", harness.HEADLINE)]
dt = harness.torch_dtype(w.dtype)
shapes, lsi = harness.level_tensors(w, "cuda")
s, c = w.positions, w.heads * w.head_dim
layer = MSDeformAttn(c, w.n_levels, w.heads, w.points).cuda().to(dt)
torch.nn.init.normal_(layer.sampling_offsets.weight, std=0.01)
torch.nn.init.normal_(layer.attention_weights.weight, std=0.05)
feats = torch.randn(w.batch, s, c, device="cuda", dtype=dt)
ref = harness.encoder_reference_points(w, "cuda").to(dt)[None, :, None, :].expand(w.batch, s, w.n_levels, 2).contiguous()
g = torch.randn(w.batch, s, c, device="cuda", dtype=dt)
def step():
    x = feats.detach().requires_grad_(True)
    layer(x, ref, x, shapes, lsi).backward(g)
def timeit(n=10):
    for _ in range(3): step()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for fused in (True, False, True, False):
    layer.fused_points = fused
    print(f"{w.name}: layer fwd+bwd, fused_points={fused}: {timeit():.3f} ms", flush=True)

# the encoder stack (LAYERS layers of {MSDeformAttn, LayerNorm, FFN, LayerNorm}) on the same input
if os.environ.get("LAYERS"):
    from unipercept_amd.deform_attn import MSDeformAttnEncoder
    n_layers = int(os.environ["LAYERS"])
    enc = MSDeformAttnEncoder(n_layers, d_model=c, d_ffn=1024, n_levels=w.n_levels, n_heads=w.heads,
                              n_points=w.points).cuda().to(dt)
    levels = [list(x) for x in w.levels]
    def step():
        x = feats.detach().requires_grad_(True)
        enc(x, shapes, lsi, spatial_shapes_list=levels).backward(g)
    print(f"{w.name}: {n_layers}-layer encoder fwd+bwd: {timeit(5):.3f} ms "
          f"({sum(p.numel() for p in enc.parameters())} parameters)", flush=True)
