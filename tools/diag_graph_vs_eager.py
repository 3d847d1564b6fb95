"""Diagnostic: run-to-run and eager-vs-graph differences of the Arch B step (64x64, B=2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import usseg_oracle as O
from test_gpu_step import _arch_b, _batches

bs = _batches(3, 2, 64, 64)
a, b, g = _arch_b(), _arch_b(), _arch_b()
g.capture_graph(bs[0][0], bs[0][1].float())
for i, (x, y) in enumerate(bs):
    la, _ = a.train_step(x, y.float()); ga = a.flat.grad.clone(); sa = a.optimizer.sumsq[0].clone()
    lb, _ = b.train_step(x, y.float()); gb = b.flat.grad.clone(); sb = b.optimizer.sumsq[0].clone()
    lg, _ = g.train_step(x, y.float()); gg = g.flat.grad.clone(); sg = g.optimizer.sumsq[0].clone()
    torch.cuda.synchronize()
    print(f"step {i}: loss eager {la.item():.6f} eager2 {lb.item():.6f} graph {lg.item():.6f}")
    print(f"   sumsq {sa.item():.8e} {sb.item():.8e} {sg.item():.8e}")
    for nm, t in (("eager2", gb), ("graph", gg)):
        d = (t.double() - ga.double())
        print(f"   grad {nm} vs eager: rel {d.norm().item() / ga.double().norm().item():.3e} nonzero diffs {(d != 0).sum().item()}")
    for nm, n in (("eager2", b), ("graph", g)):
        d = (n.flat.flat.double() - a.flat.flat.double())
        print(f"   params {nm} vs eager: max abs {d.abs().max().item():.3e} nonzero {(d != 0).sum().item()}")
    # which tensors differ in grad (graph)
    if i == 0:
        for (k, p), (_, q) in zip(a.named_parameters(), g.named_parameters()):
            dd = (p.grad.double() - q.grad.double()).abs().max().item()
            if dd > 0:
                print(f"      {k}: max abs grad diff {dd:.3e} (|g| max {p.grad.abs().max().item():.3e})")
