"""Eager conv launches for a rocprofv3 --pmc pass: python tools/pmc_conv.py B H Cin Cout d [reps]"""
import sys, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
dev = torch.device("cuda:0")
B, H, Cin, Cout, d = [int(a) for a in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
wf = (torch.randn(max(16, (Cout + 15) // 16 * 16), 9 * Cin, device=dev) * 0.02).to(torch.bfloat16)
bias = torch.zeros(Cout, device=dev)
y = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.conv2d_fwd(x, wf, bias, 3, d, y)
torch.cuda.synchronize()
