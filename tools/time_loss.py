import sys, torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
B, H, W, C = 16, 256, 256, 3
logits = torch.randn(B, H // 2, W // 2, 16, device="cuda")
y = torch.softmax(torch.randn(B, H, W, C, device="cuda"), -1)
probs = torch.empty(B, H, W, C, device="cuda")
loss = torch.zeros(ops.ACC_FLOATS, device="cuda")
dl = torch.empty(B, H // 2, W // 2, 16, dtype=torch.bfloat16, device="cuda")
big = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
def run(): ops.softmax_loss(logits, y, probs, loss, dl, HW=H * W, C_classes=C, loss_kind=0, inv_global_batch=1 / 16, quad_w=W)
for _ in range(3): run()
for flush in (False, True):
    ts = []
    for _ in range(10):
        if flush: big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("flush" if flush else "hot", [f"{t:.1f}" for t in ts])
