"""With learning rate 0 every step has the same inputs: gradients must repeat bit for bit, also while an unrelated stream keeps the GPU busy.
Lists the parameters whose gradients differ between steps: python tools/diag_load_race.py B HW steps load(0|1) graph(0|1)"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import usseg_oracle as O
from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
B, HW, n, load, graph = (int(a) for a in sys.argv[1:6])
arch = sys.argv[6] if len(sys.argv) > 6 else "B"
x, y = O.synthetic_batch(B, HW, HW, 1, seed=40)
x, y = x.cuda(), y.float().cuda()
if arch == "B":
    net = VisionTransformer(batch_size=B, img_size=(HW, HW), in_channels=1, device="cuda:0", seed=0, learning_rate=0.0)
elif arch == "T":
    from ultrasound_modeling_amd.TBI_TransUNet import VisionTransformer as TransUNet
    net = TransUNet(img_size=(HW, HW), batch_size=B, in_channels=1, device="cuda:0", seed=0, learning_rate=0.0)
elif arch == "A":
    from ultrasound_modeling_amd.TBI_ResNest import ResNest
    net = ResNest(HW, HW, 1, 3, ksize=3, radix=3, kpaths=4, learning_rate=0.0, device="cuda:0", seed=0)
    net.train_step = lambda a, b: net.step(a, b, train=True)[::2]
    net.named_parameters = net.resModel.named_parameters
    net.resModel.injected_masks = [None] * 8          # (the always-on dropout draws a fresh mask every step: switched off for the repeat test)
else:
    from ultrasound_modeling_amd.SwinTransformer import SwinTransformerModel
    net = SwinTransformerModel(model_name="s", img_size=(HW, HW), patch_size=(4, 4), in_chans=1, embed_dim=96, depths=[2, 2, 6, 2],
                               num_heads=[3, 6, 12, 24], window_size=8, device="cuda:0", seed=0, learning_rate=0.0)
    y = torch.full((B, 768), 1.0 / 768, device="cuda")
if graph:
    net.capture_graph(x, y)
side = torch.cuda.Stream()
a = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
b = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
big = torch.empty(64 << 20, device="cuda")
ref, bad, nbad = None, {}, 0
for i in range(n):
    if load:
        with torch.cuda.stream(side):
            for _ in range(6):
                c = a @ b
                big.add_(1.0)
    l, p = net.train_step(x, y)
    torch.cuda.synchronize()
    g = net.flat.grad.clone()
    if ref is None:
        ref, lref, pref = g, l.clone(), p.clone()
        continue
    if not torch.equal(g, ref):
        nbad += 1
        d = (g != ref)
        for name, q in net.named_parameters():
            o = (q.grad.data_ptr() - net.flat.grad.data_ptr()) // 4
            k = int(d[o:o + q.numel()].sum().item())
            if k:
                bad.setdefault(name, []).append((i, k, q.numel()))
    if not torch.equal(p, pref):
        print("step", i, "outputs differ:", int((p != pref).sum().item()), "loss", l.sum().item(), lref.sum().item())
print(f"{nbad} of {n - 1} steps differ;", len(bad), "parameters affected")
mods = {}
for k in bad:
    key = ".".join(k.split(".")[:4])
    mods[key] = mods.get(key, 0) + 1
for k, v in sorted(mods.items()):
    print("  ", v, k)
