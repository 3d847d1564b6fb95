"""Time the fused cardinal-group forward launch (csrc/cardinal.hip) per residual_S stage at the bench's sizes, HIP-event timed in a loop.
python tools/time_cardinal.py [B=16] [HW=256] [--phases]
--phases: builds csrc/cardinal.hip with -DCARD_TIMING into gpurun_out/libcard_timing.so (workgroup (0,0,z) stamps the 100 MHz wall clock at
its phase boundaries) and prints the phase times of one workgroup per stage."""
import ctypes, os, subprocess, sys
PH = "--phases" in sys.argv
sys.argv = [a for a in sys.argv if a != "--phases"]
if PH:
    cs, out = "ultrasound_modeling_amd/csrc", "gpurun_out/libcard_timing.so"
    os.makedirs("gpurun_out", exist_ok=True)
    objs = [os.path.join(cs, f) for f in os.listdir(cs) if f.endswith(".o") and f != "cardinal.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Iinclude", "-I" + cs, "-DCARD_TIMING", "-fno-slp-vectorize",
                    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-c", cs + "/cardinal.hip", "-o", "gpurun_out/card_timing.o"], check=True,
                   stderr=subprocess.DEVNULL)
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "gpurun_out/card_timing.o"] + objs + ["-o", out], check=True)
    os.environ["USSEG_LIB"] = os.path.abspath(out)
import torch
sys.path.insert(0, ".")
from ultrasound_modeling_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda"
for st, (cin, cv11, cvkk, oc) in enumerate(((32, 3, 10, 64), (64, 7, 21, 128), (128, 14, 42, 256), (256, 28, 85, 512))):
    h = HW // (2 << st)
    P = 3
    Up, Vp = (P * cv11 + 7) // 8 * 8, (P * cvkk + 7) // 8 * 8
    r16 = lambda n: (n + 15) // 16 * 16
    x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
    w1 = (torch.randn(r16(Up), cin, device=dev) * 0.1).to(torch.bfloat16)
    w2 = (torch.randn(r16(Vp), 9 * Up, device=dev) * 0.1).to(torch.bfloat16)
    wsc = (torch.randn(oc, cin, device=dev) * 0.1).to(torch.bfloat16)
    f = lambda n: torch.randn(n, device=dev) * 0.1
    args = (x, w1, f(Up), 1 + f(Up), f(Up), w2, f(Vp), 1 + f(Vp), f(Vp), wsc, f(oc), 1 + f(oc), f(oc), P, cv11, cvkk, Up, Vp, oc, 1e-3, 0.3)
    for _ in range(3):
        ops.cardinal_fwd(*args)
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.cardinal_fwd(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f"stage {st + 1}: {h}x{h} cin {cin}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per launch")
    if PH:
        from ultrasound_modeling_amd import _lib
        buf = (ctypes.c_ulonglong * 32)()
        _lib.load().usseg_cardinal_debug_read(buf)
        card, sc = list(buf[:10]), list(buf[16:22])
        names = ["x tile load", "GEMM1", "LN1", "u copy-out", "GEMM2 loop", "GEMM2 epilogue", "v_raw copy-out", "LN2", "y copy-out + pool"]
        print("   cardinal workgroup (us):", ", ".join(f"{n} {(b_ - a_) / 100:.2f}" for n, a_, b_ in zip(names, card, card[1:])), f"| total {(card[9] - card[0]) / 100:.2f}")
        names = ["x tile load", "GEMM", "sc_raw copy-out", "LN", "sc copy-out"]
        print("   shortcut workgroup (us):", ", ".join(f"{n} {(b_ - a_) / 100:.2f}" for n, a_, b_ in zip(names, sc, sc[1:])), f"| total {(sc[5] - sc[0]) / 100:.2f}")

# ---- backward: the fused launch against the four launches it replaces
print("backward:")
for st, (cin, cv11, cvkk, oc) in enumerate(((32, 3, 10, 64), (64, 7, 21, 128), (128, 14, 42, 256), (256, 28, 85, 512))):
    h = HW // (2 << st)
    P = 3
    U, V = P * cv11, P * cvkk
    Up, Vp = (U + 7) // 8 * 8, (V + 7) // 8 * 8
    r16 = lambda n: (n + 15) // 16 * 16
    t = lambda c: torch.randn(B, h, h, c, device=dev).to(torch.bfloat16)
    v_raw, dout, u_raw, sc_raw, dsc = t(Vp), t(Vp), t(Up), t(oc), t(oc)
    w2d = (torch.randn(r16(Up), 9 * Vp, device=dev) * 0.1).to(torch.bfloat16)
    f = lambda n: torch.randn(n, device=dev) * 0.1
    g2, be2, g1, be1, gsc, besc = 1 + f(Vp), f(Vp), 1 + f(Up), f(Up), 1 + f(oc), f(oc)
    sa_s, sa_dg = torch.rand(B, V, device=dev), f(B * V).reshape(B, V)
    z = lambda n: torch.zeros(n, device=dev)
    gr = [z(Vp), z(Vp), z(Vp), z(Up), z(Up), z(Up), z(oc), z(oc), z(oc)]
    dv, dcat = ops.new_act(B, h, h, Vp, dev), ops.new_act(B, h, h, Up + oc, dev)
    du = torch.empty_like(u_raw)

    def fused():
        ops.cardinal_bwd(dout, dsc, v_raw, u_raw, sc_raw, w2d, g2, be2, g1, be1, gsc, besc, sa_s, sa_dg, 3.0, dv, dcat, gr, cin, P, cv11, cvkk, Up, Vp, oc, 1e-3, 0.3)

    def unfused():
        ops.norm_act_bwd_sa(v_raw, dout, V, g2, be2, dv, gr[0], gr[1], 0, P, 1e-3, ops.ACT_LRELU, 0.3, sa_s, sa_dg, 3.0, dbias=gr[2])
        ops.conv2d_dgrad(dv, w2d, 3, 1, du)
        ops.norm_act_bwd(u_raw, du, U, g1, be1, dcat[..., :Up], gr[3], gr[4], 0, P, 1e-3, ops.ACT_LRELU, 0.3, dbias=gr[5])
        ops.norm_act_bwd(sc_raw, dsc, oc, gsc, besc, dcat[..., Up:], gr[6], gr[7], 0, 1, 1e-3, ops.ACT_LRELU, 0.3, dbias=gr[8])
    res = []
    for fn in (fused, unfused):
        for _ in range(3):
            fn()
        n = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / n)
    print(f"stage {st + 1}: {h}x{h}: fused {res[0]:.1f} us, the four launches + their finishing reductions {res[1]:.1f} us")
    if PH:
        fused()
        torch.cuda.synchronize()
        from ultrasound_modeling_amd import _lib
        buf = (ctypes.c_ulonglong * 32)()
        _lib.load().usseg_cardinal_debug_read(buf)
        card = list(buf[:9])
        names = ["tile loads", "LN2 row passes", "LN2 column pass", "dv copy-out", "3x3 dgrad GEMM", "LN1 row pass", "LN1 column pass", "du_raw copy-out"]
        print("   cardinal workgroup 0, its last tile (us):", ", ".join(f"{n} {(b_ - a_) / 100:.2f}" for n, a_, b_ in zip(names, card, card[1:])),
              f"| total {(card[8] - card[0]) / 100:.2f}")
