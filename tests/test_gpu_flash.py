"""GPU parity of the fused attention kernels (csrc/flash_attn.hip; VisionTransformer.py:38-50, TBI_TransUNet.py:44-62):
O = softmax(q k^T / sqrt(num_heads)) v per (image, head) and its three gradients, against an fp64 PyTorch restatement with autograd
on the SAME bf16-representable inputs.  The kernel rounds the probabilities and dS to bf16 before the second GEMMs (as the unfused path
does): bars 2e-3 (forward, output stored bf16) and 6e-3 (gradients, stored bf16); also bit-reproducibility and agreement with the
unfused launches (batched GEMMs + row softmax)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def reference(qkv, d_out, nh, scale):
    B, N, _, C3 = qkv.shape
    hs = C3 // 3
    x = qkv.double().reshape(B, N, 3, nh, hs // nh).requires_grad_(True)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))                  # [B, nh, N, dh]   (:38-40 split_heads)
    p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)                     # :41-43
    o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, 1, hs)                           # :47-49
    (g,) = torch.autograd.grad(o, x, d_out.double())
    return o.detach(), g.reshape(B, N, 1, C3), p.detach()


def unfused(qkv, d_out, nh, scale):
    """The launches Attention.forward/backward use when the weights are requested: batched GEMMs + row softmax (bf16 P, bf16 dS)."""
    from ultrasound_modeling_amd import ops
    B, N, _, C3 = qkv.shape
    hs, dev, BF = C3 // 3, qkv.device, torch.bfloat16
    dh = hs // nh
    q, k, v = qkv[..., :hs], qkv[..., hs:2 * hs], qkv[..., 2 * hs:]
    s_pp, s_qkv, s_ctx, s_hd = (nh * N * N, N * N), (N * 3 * hs, dh), (N * hs, dh), (nh * N * dh, N * dh)
    S = torch.empty((B, nh, N, N), dtype=torch.float32, device=dev)
    ops.gemm_nt_batched(q, k, S, N, N, dh, 3 * hs, 3 * hs, N, B, nh, s_qkv, s_qkv, s_pp, out_f32=True)
    P32, Pb = torch.empty_like(S), torch.empty((B, nh, N, N), dtype=BF, device=dev)
    ops.softmax_rows_fwd(S, N, scale, P32, Pb)
    vt, kt = torch.empty((B * nh, dh, N), dtype=BF, device=dev), torch.empty((B * nh, dh, N), dtype=BF, device=dev)
    ops.transpose_batched(v, N, dh, 3 * hs, B, nh, s_qkv, vt)
    ops.transpose_batched(k, N, dh, 3 * hs, B, nh, s_qkv, kt)
    ctx = ops.new_act(B, N, 1, hs, dev)
    ops.gemm_nt_batched(Pb, vt, ctx, N, dh, N, N, N, hs, B, nh, s_pp, (nh * dh * N, dh * N), s_ctx)
    dqkv = ops.new_act(B, N, 1, 3 * hs, dev)
    dV = torch.zeros((B, nh, N, dh), dtype=torch.float32, device=dev)
    ops.gemm_tn_batched(Pb, d_out, dV, N, dh, N, N, hs, B, nh, s_pp, s_ctx, s_hd)
    dP = torch.empty((B, nh, N, N), dtype=torch.float32, device=dev)
    ops.gemm_nt_batched(d_out, v, dP, N, N, dh, hs, 3 * hs, N, B, nh, s_ctx, s_qkv, s_pp, out_f32=True)
    dS = torch.empty((B, nh, N, N), dtype=BF, device=dev)
    ops.softmax_rows_bwd(P32, dP, N, scale, dS)
    ops.gemm_nt_batched(dS, kt, dqkv, N, dh, N, N, N, 3 * hs, B, nh, s_pp, (nh * dh * N, dh * N), s_qkv)
    dK = torch.zeros((B, nh, N, dh), dtype=torch.float32, device=dev)
    ops.gemm_tn_batched(dS, q, dK, N, dh, N, N, 3 * hs, B, nh, s_pp, s_qkv, s_hd)
    ops.cast_f32_to_bf16_batched(dK, N, dh, B, nh, dqkv[..., hs:2 * hs], 3 * hs, s_qkv)
    ops.cast_f32_to_bf16_batched(dV, N, dh, B, nh, dqkv[..., 2 * hs:], 3 * hs, s_qkv)
    return ctx, dqkv


@pytest.mark.parametrize("with_o32", [True, False])
# (8, 1024, 4) and (16, 2048, 1) fill the chip with 128-query workgroups: the 8-wave tiling; the others run the 4-wave one
@pytest.mark.parametrize("B,N,nh", [(2, 32, 4), (1, 48, 2), (2, 200, 4), (2, 256, 4), (1, 1024, 2), (3, 129, 1), (8, 1024, 4), (16, 2000, 1)])
def test_flash_attention_matches_fp64(B, N, nh, with_o32):
    from ultrasound_modeling_amd import ops
    torch.manual_seed(N + nh)
    hs = 128 * nh
    scale = 1.0 / math.sqrt(nh)
    qkv = (torch.randn(B, N, 1, 3 * hs) * 0.8).to(torch.bfloat16).to(DEV)
    d_out = torch.randn(B, N, 1, hs).to(torch.bfloat16).to(DEV)
    out = torch.full((B, N, 1, hs), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.empty((B * nh, N), dtype=torch.float32, device=DEV)
    o32 = torch.full((B, N, hs), float("nan"), dtype=torch.float32, device=DEV) if with_o32 else None
    ops.flash_attn_fwd(qkv, nh, scale, out, lse, o32)
    o_ref, g_ref, p_ref = reference(qkv.cpu(), d_out.cpu(), nh, scale)
    e_o = rel(out, o_ref)
    # base-2 log-sum-exp of the scaled scores
    x = qkv.cpu().double().reshape(B, N, 3, nh, 128)
    s = (x[:, :, 0].permute(0, 2, 1, 3) @ x[:, :, 1].permute(0, 2, 3, 1)) * scale
    lse_ref = (torch.logsumexp(s, dim=-1) / math.log(2.0)).reshape(B * nh, N)
    e_l = (lse.double().cpu() - lse_ref).abs().max().item()
    dqkv = torch.full((B, N, 1, 3 * hs), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.empty((B * nh, N), dtype=torch.float32, device=DEV)
    ops.flash_attn_bwd(qkv, nh, scale, out, d_out, lse, delta, dqkv, o32)
    torch.cuda.synchronize()
    if with_o32:
        assert rel(o32.reshape(B, N, 1, hs), o_ref) < 5e-5 and torch.equal(o32.reshape(B, N, 1, hs).to(torch.bfloat16), out)
        # rows of dS sum to zero: the key gradient summed over keys (= the key bias gradient) vanishes
        kb = dqkv[..., hs:2 * hs].float().sum((0, 1, 2))
        qb = dqkv[..., :hs].float().sum((0, 1, 2))
    if with_o32 and N % 8 == 0:          # the batched-GEMM path needs 8-element aligned score rows
        ctx_u, dqkv_u = unfused(qkv, d_out, nh, scale)
        kb_u = dqkv_u[..., hs:2 * hs].float().sum((0, 1, 2))
        e_u = [rel(dqkv_u[..., i * hs:(i + 1) * hs], g_ref[..., i * hs:(i + 1) * hs]) for i in range(3)]
        print(f"   key-bias gradient max {kb.abs().max().item():.3e} (unfused launches {kb_u.abs().max().item():.3e}) vs query-bias gradient max "
              f"{qb.abs().max().item():.3e}; unfused vs fp64: out {rel(ctx_u, o_ref):.2e} dq {e_u[0]:.2e} dk {e_u[1]:.2e} dv {e_u[2]:.2e}")
        assert rel(out, ctx_u) < 3e-3 and rel(dqkv, dqkv_u) < 8e-3
        assert kb.abs().max().item() < 3 * kb_u.abs().max().item() + 1e-3 * qb.abs().max().item()
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
    e_q, e_k, e_v = (rel(dqkv[..., i * hs:(i + 1) * hs], g_ref[..., i * hs:(i + 1) * hs]) for i in range(3))
    print(f"B={B} N={N} heads={nh}: out {e_o:.2e} lse {e_l:.2e} dq {e_q:.2e} dk {e_k:.2e} dv {e_v:.2e}")
    assert e_o < 2e-3 and e_l < 2e-3       # output stored bf16: 2^-9 / sqrt(3) = 1.1e-3 RMS per element
    assert max(e_q, e_k, e_v) < (4e-3 if with_o32 else 6e-3)   # without the fp32 copy of O, delta carries O's bf16 rounding
    # bitwise reproducible
    out2, dqkv2 = torch.empty_like(out), torch.empty_like(dqkv)
    ops.flash_attn_fwd(qkv, nh, scale, out2, lse, o32)
    ops.flash_attn_bwd(qkv, nh, scale, out2, d_out, lse, delta, dqkv2, o32)
    assert torch.equal(out, out2) and torch.equal(dqkv, dqkv2)


def test_flash_attention_rejects_other_head_sizes():
    from ultrasound_modeling_amd import ops
    qkv = torch.zeros(1, 16, 1, 3 * 64, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(1, 16, 1, 64, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(1, 16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.flash_attn_fwd(qkv, 1, 1.0, out, lse)
