"""Finds the first forward launch whose output changes under foreign GPU load: every wrapped op's output is cloned (asynchronously), a quiet
pass gives the references, loaded passes are compared with them.  python tools/diag_load_first.py B HW passes"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import usseg_oracle as O
from ultrasound_modeling_amd import ops
from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
B, HW, n = (int(a) for a in sys.argv[1:4])
net = VisionTransformer(batch_size=B, img_size=(HW, HW), in_channels=1, device="cuda:0", seed=0, learning_rate=0.0)
x, y = O.synthetic_batch(B, HW, HW, 1, seed=40)
x, y = x.cuda(), y.float().cuda()
rec = []
def wrap(name, out_of):
    f = getattr(ops, name)
    def g(*a, **k):
        r = f(*a, **k)
        for t in out_of(a, k, r):
            if torch.is_tensor(t):
                rec.append((name, tuple(t.shape), t.clone()))
        return r
    setattr(ops, name, g)
wrap("cast_input", lambda a, k, r: [r])
wrap("conv2d_fwd", lambda a, k, r: [a[5]])
wrap("conv2d_fwd_multi", lambda a, k, r: [j[5] for j in a[0]])
wrap("tconv2d_fwd", lambda a, k, r: [a[4]])
wrap("norm_act_fwd", lambda a, k, r: [a[4]])
wrap("norm_act_fwd_gap", lambda a, k, r: [a[4], r[1][0]])
wrap("bn_act_pool_fwd", lambda a, k, r: [r])
wrap("avgpool2_fwd", lambda a, k, r: [a[1]])
wrap("splitattn_fwd", lambda a, k, r: [a[3]])
wrap("reinject_hidden", lambda a, k, r: list(a[1]))
side = torch.cuda.Stream()
A_ = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
B_ = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
big = torch.empty(64 << 20, device="cuda")
net.step(x, y); torch.cuda.synchronize()
rec.clear()
net.step(x, y); torch.cuda.synchronize()
ref = list(rec)
print(len(ref), "recorded outputs per pass")
first = {}
for it in range(n):
    rec.clear()
    with torch.cuda.stream(side):
        for _ in range(8):
            c = A_ @ B_
            big.add_(1.0)
    net.step(x, y)
    torch.cuda.synchronize()
    for i, ((nm, sh, t), (_, _, r)) in enumerate(zip(rec, ref)):
        if not torch.equal(t, r):
            d = (t.float() - r.float()).abs()
            key = (i, nm, sh)
            first[key] = first.get(key, 0) + 1
            print(f"pass {it}: first difference at output {i} {nm} {sh}: {int((d > 0).sum())} elements, max abs {d.max().item():.3e}")
            break
print("summary:", first)
