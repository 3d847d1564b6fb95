"""One grouped per-pixel LayerNorm launch repeated under foreign GPU load; compares every output with the quiet reference.
python tools/diag_norm_load.py B H W C G reps [both|gemm|stream|none]"""
import sys, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
B, H, W, C, G, reps = (int(a) for a in sys.argv[1:7])
load = sys.argv[7] if len(sys.argv) > 7 else "both"
Cp = (C + 7) // 8 * 8
torch.manual_seed(0)
x = torch.zeros(B, H, W, Cp, dtype=torch.bfloat16, device="cuda")
x[..., :C] = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
gamma = (1 + 0.1 * torch.randn(C, device="cuda")).float()
beta = (0.1 * torch.randn(C, device="cuda")).float()
def run():
    return ops.norm_act_fwd(x, C, gamma, beta, torch.empty_like(x), 0, G, 1e-3, ops.ACT_LRELU, 0.3)
ref = run().clone(); torch.cuda.synchronize()
side = torch.cuda.Stream()
A_ = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
big = torch.empty(64 << 20, device="cuda")
bad = 0
for it in range(reps):
    with torch.cuda.stream(side):
        for _ in range(4):
            if load in ("both", "gemm"):
                c = A_ @ A_
            if load in ("both", "stream"):
                big.add_(1.0)
    outs = [run() for _ in range(8)]
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o, ref):
            bad += 1
            d = (o.float() - ref.float()).abs()
            idx = (d > 0).nonzero()
            if bad <= 3 and len(sys.argv) > 8:      # verbose: the wrong pixels, their position in the launch and the values
                pix = idx[:, :3].unique(dim=0)
                flat = (pix[:, 0] * H + pix[:, 1]) * W + pix[:, 2]
                print("  flat pixel indices", flat[:24].tolist(), " (mod 64:", sorted(set((flat % 64).tolist()))[:16], ")")
                b_, h_, w_ = pix[0].tolist()
                print("  x    ", [round(v, 3) for v in x[b_, h_, w_].float().tolist()])
                print("  ref  ", [round(v, 3) for v in ref[b_, h_, w_].float().tolist()])
                print("  bad  ", [round(v, 3) for v in o[b_, h_, w_].float().tolist()])
            if bad <= 3:
                print("mismatch:", int((d > 0).sum()), "elements; pixels", idx[:, :3].unique(dim=0).shape[0], "first", idx[0].tolist(), "channels", sorted(set(idx[:, 3].tolist()))[:16])
print(f"{bad} wrong of {reps * 8} launches")
