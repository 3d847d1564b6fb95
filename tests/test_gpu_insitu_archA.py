"""Per-layer IN-SITU parity of Arch A (GPU): north_star's "per-layer fwd/bwd within 1e-3 rel of reference on identical inputs"
for the model of TBI_ResNest.py:80-220 + my_loss_cat (:234-248) - the DP=8 headline model of BASELINE configs[2].

Same method as tests/test_gpu_insitu.py (Arch B): the whole network (64x64x1, B=2, dropout masks injected) runs ONCE in the oracle
with bf16 storage emulation and a recorder around its primitives; every launch of the product's forward and backward pass is then
replayed one at a time, through the model's own packed operands and flat-buffer views, on the oracle's tensors for that layer and
compared with that layer recomputed locally in fp64:

    bf16 outputs (activations, input gradients)  <= 1e-3 relative L2 against the fp64 result rounded to bf16
    fp32 outputs (weight / bias / norm-parameter gradients, probabilities, loss map)  <= 1e-3 against fp64

Covered, per stage (5 stages; the two 512-wide ones run as two slabs of two paths): the radix x paths 1x1 convs as ONE GEMM
(:162), BatchNorm + ELU over all branches (:164-165), the block-diagonal 3x3 (:167), BatchNorm + ELU with the pooled partial rows
(:169-170,:186), the split attention with R=3 DISTINCT branches, BatchNorm-mode MLP and one dense2 per radix (:175-207) forward and
backward, concats_2 with the residual epilogue (:140,:148), the shortcut conv + BN + ELU where it exists (:142-145); the stem
(:83-92), the six pools, upsample_0..4 = 4x4 transposed conv + BN + always-on dropout + ReLU (:209-220), f_tran in quad form +
softmax + my_loss_cat (:124-125,:234-248).
"""
import pytest
import torch

import usseg_oracle as O
from test_gpu_insitu import TOL, bf, check, dev, fresh, g_of

pytestmark = pytest.mark.gpu
DEV = "cuda"
R, KP = 3, 4


def ste(t):
    """bf16 storage of an intermediate inside a fused launch (value and gradient), as the oracle's STORAGE_DTYPE does."""
    return O._RoundSTE.apply(t, torch.bfloat16, True)


class RecorderA:
    NAMES = ("conv2d_same", "conv2d_transpose_s2_same", "batch_norm", "elu", "avg_pool2", "archA_split_attention", "archA_upsample",
             "archA_residual_S")

    def __init__(self, P):
        self.id2name = {id(v): k for k, v in P.items()}
        self.conv, self.norm, self.sa, self.pool, self.acts, self.up, self.stage = {}, {}, {}, [], [], {}, {}
        self.orig = {n: getattr(O, n) for n in self.NAMES}
        self.in_sa = 0

    @staticmethod
    def _keep(t):
        if t.requires_grad:
            t.retain_grad()
        return t

    def __enter__(self):
        o, rec = self.orig, self

        def conv2d_same(x, w, b=None, dilation=1):
            y = o["conv2d_same"](x, w, b, dilation)
            if not rec.in_sa:
                rec.conv[rec.id2name[id(w)][:-len(".kernel")]] = (x, rec._keep(y))
            return y

        def conv2d_transpose_s2_same(x, w, b=None):
            y = o["conv2d_transpose_s2_same"](x, w, b)
            rec.conv[rec.id2name[id(w)][:-len(".kernel")]] = (x, rec._keep(y))
            return y

        def batch_norm(x, gamma, beta, mm, mv, training=None):
            y = o["batch_norm"](x, gamma, beta, mm, mv, training)
            if not rec.in_sa:
                rec.norm[rec.id2name[id(gamma)][:-len(".gamma")]] = (x, y)
            return y

        def elu(x):
            y = o["elu"](x)
            if not rec.in_sa:
                rec.acts.append((x, rec._keep(y)))
            return y

        def avg_pool2(x):
            y = o["avg_pool2"](x)
            rec.pool.append((x, rec._keep(y)))
            return y

        def archA_split_attention(inputs, P, prefix):
            rec.in_sa += 1
            try:
                y = o["archA_split_attention"](inputs, P, prefix)
            finally:
                rec.in_sa -= 1
            rec.sa[prefix] = (list(inputs), rec._keep(y))
            return y

        def archA_upsample(x, P, name, dropout_mask):
            y = o["archA_upsample"](x, P, name, dropout_mask)
            rec.up[name] = (x, rec._keep(y), dropout_mask)
            return y

        def archA_residual_S(x, P, name, radix, kpaths):
            y = o["archA_residual_S"](x, P, name, radix, kpaths)
            rec.stage[name] = (x, rec._keep(y))
            return y
        for n, f in zip(self.NAMES, (conv2d_same, conv2d_transpose_s2_same, batch_norm, elu, avg_pool2, archA_split_attention, archA_upsample,
                                     archA_residual_S)):
            setattr(O, n, f)
        return self

    def __exit__(self, *a):
        for n, f in self.orig.items():
            setattr(O, n, f)

    def act_after(self, x):
        for xi, yi in self.acts:
            if xi is x:
                return yi
        raise KeyError("no ELU recorded for this tensor")


@pytest.fixture(scope="module")
def world():
    from ultrasound_modeling_amd.TBI_ResNest import ResNest
    P = O.init_archA_params(channel=1, radix=R, kpaths=KP, seed=21, perturb=True)
    P = {k: (bf(v) if k.endswith(".kernel") and v.dim() == 4 and "_att" not in k else v.float().double()) for k, v in P.items()}
    net = ResNest(64, 64, 1, 3, ksize=3, radix=R, kpaths=KP, learning_rate=5e-3)
    net.load_params(P)
    gen = torch.Generator().manual_seed(22)
    keep = [(torch.rand(2, 2 ** (i + 1), 2 ** (i + 1), 512, generator=gen) > 0.5).double() for i in range(3)]
    x, y = O.synthetic_batch(2, 64, 64, 1, seed=23)
    names = O.trainable_names(P)
    leaves = {n: P[n].clone().requires_grad_(True) for n in names}
    Pl = dict(P)
    Pl.update(leaves)
    rec = RecorderA(Pl)
    O.STORAGE_DTYPE = torch.bfloat16
    try:
        with rec:
            probs = O.archA_forward(bf(x), Pl, R, KP, dropout_masks=keep)
            O.my_loss_cat(y, probs, 64, 64).sum().backward()
    finally:
        O.STORAGE_DTYPE = None
    rec.P, rec.x, rec.y, rec.keep = P, bf(x), y, keep
    return net, P, rec


def bn_ref(x, P, name, g=None, b=None):
    return O.batch_norm(x, P[name + ".gamma"] if g is None else g, P[name + ".beta"] if b is None else b, P[name + ".moving_mean"],
                        P[name + ".moving_variance"])


def cat_dev(ts, width=None):
    return dev(torch.cat([t.detach() for t in ts], dim=3), width)


def conv_check(net, rec, P, name, layer, worst, residual=None, transposed=False, need_dx=True, skip_bias=False):
    """One Conv2D / Conv2DTranspose on the oracle's input: forward (+ residual epilogue), backward-data, weight (and bias) gradient."""
    x, y = rec.conv[name]
    x, dy = x.detach(), g_of(y)
    xl, wl, bl = fresh(x, P[name + ".kernel"], P[name + ".bias"])
    ref = (O.conv2d_transpose_s2_same if transposed else O.conv2d_same)(xl, wl, bl)
    out_ref = ref if residual is None else ref + residual
    kw = {} if residual is None else {"residual": dev(residual)}
    out = layer.forward(dev(x, layer.cin_p), **kw)
    check(name + " fwd", out[..., :layer.cout], bf(out_ref), worst=worst)
    if layer.cout_p > layer.cout:
        assert out[..., layer.cout:].abs().max().item() == 0, name + ": pad channels"
    gx, gw, gb = torch.autograd.grad(ref, [xl, wl, bl], dy)
    net.flat.zero_grad()
    if transposed:
        dx = layer.backward(dev(dy, layer.cout_p), need_dx=need_dx)
    else:
        dx = layer.backward(dev(dy, layer.cout_p), need_dx=need_dx, skip_bias=skip_bias)
    torch.cuda.synchronize()
    if need_dx:
        check(name + " dgrad", dx[..., :layer.cin], bf(gx), worst=worst)
    check(name + " wgrad", layer.kernel.grad, gw, worst=worst)
    if not skip_bias:
        check(name + " dbias", layer.bias.grad, gb, worst=worst)


def elu_check(rec, y_raw, conv, worst, tag):
    """ELU on a stored conv output (:84,:87) and its backward fused with the conv's bias gradient (column sums of the result)."""
    from ultrasound_modeling_amd import ops
    out_rec = rec.act_after(y_raw)
    y_raw = y_raw.detach()
    yl, = fresh(y_raw)
    ref = O.elu(yl)
    rd = dev(y_raw)
    t = ops.act_fwd(rd, torch.empty_like(rd), ops.ACT_ELU, 1.0)
    check(tag + " ELU fwd", t[..., :conv.cout], bf(ref), worst=worst)
    dy = g_of(out_rec)
    gy, = torch.autograd.grad(ref, yl, dy)
    db = torch.zeros(conv.cout_p, dtype=torch.float32, device=DEV)
    d = ops.act_bwd_colsum(rd, dev(dy), torch.empty_like(rd), ops.ACT_ELU, 1.0, db, conv.cout)
    torch.cuda.synchronize()
    check(tag + " ELU bwd", d[..., :conv.cout], bf(gy), worst=worst)
    check(tag + " bias grad (column sums)", db[:conv.cout], bf(gy).sum(dim=(0, 1, 2)), worst=worst)


def test_stem_and_pools(world):
    from ultrasound_modeling_amd import ops
    net, P, rec = world
    m = net.resModel
    worst = []
    conv_check(net, rec, P, "Conv1", m.Conv1, worst, need_dx=False)                                 # :83
    elu_check(rec, rec.conv["Conv1"][1], m.Conv1, worst, "Conv1")                                    # :84
    conv_check(net, rec, P, "conv2_1_1", m.conv2_1_1, worst)                                        # :85
    elu_check(rec, rec.conv["conv2_1_1"][1], m.conv2_1_1, worst, "conv2_1_1")                        # :87
    conv_check(net, rec, P, "conv2_1_2", m.conv2_1_2, worst)                                        # :88
    # :90-92 BatchNorm + ELU + pool_1 as ONE launch each way (the activated 256x256 tensor is stored only in bf16 registers)
    name = "conv2_1_2bn"
    x = rec.norm[name][0].detach()
    xl, gl, bl = fresh(x, P[name + ".gamma"], P[name + ".beta"])
    ref = O.avg_pool2(ste(O.elu(bn_ref(xl, P, name, gl, bl))))
    bn = m.conv2_1_2bn
    pooled = bn.forward_pool(dev(x), ops.ACT_ELU, 1.0)
    check("stem BN+ELU+pool fwd", pooled, bf(ref), worst=worst)
    dy = g_of(rec.pool[0][1])
    gx, gg, gb = torch.autograd.grad(ref, [xl, gl, bl], dy)
    net.flat.zero_grad()
    dx = bn.backward_pool(dev(dy), dbias=m.conv2_1_2.bias.grad)
    torch.cuda.synchronize()
    check("stem BN+ELU+pool dx", dx, bf(gx), worst=worst)
    check("stem BN dgamma", bn.gamma.grad, gg, worst=worst)
    check("stem BN dbeta", bn.beta.grad, gb, worst=worst)
    check("conv2_1_2 dbias (sum dx)", m.conv2_1_2.bias.grad, gx.sum(dim=(0, 1, 2)), worst=worst)
    from ultrasound_modeling_amd.layers import AveragePooling2D
    for i in range(1, 6):                                                                            # pool_2 .. pool_6 (:95-107)
        x, y = rec.pool[i]
        xl, = fresh(x)
        ref = O.avg_pool2(xl)
        pool = AveragePooling2D()
        check(f"pool_{i + 1} fwd", pool.forward(dev(x)), bf(ref), worst=worst)
        dy = g_of(y)
        check(f"pool_{i + 1} bwd", pool.backward(dev(dy)), bf(torch.autograd.grad(ref, xl, dy)[0]), worst=worst)
    print("Arch A stem/pools worst:", [(f"{e:.2e}", n) for e, n in sorted(worst, reverse=True)[:5]])


@pytest.mark.parametrize("stage", [0, 1, 2, 3, 4])
def test_residual_S_stage_layer_by_layer(world, stage):
    from ultrasound_modeling_amd import ops
    net, P, rec = world
    m = net.resModel
    st = m._build()[stage]
    name = st.name
    worst = []
    e = 1e-3
    x = rec.stage[name][0].detach()                              # the stage input (pool_{stage+1})
    B, H, W, _ = x.shape
    c1_w = ops.roundup(st.Vtot, 8)
    for sl in st.slabs:
        tag = f"{name} paths {sl.paths[0]}-{sl.paths[-1]}: "
        brs = [f"{name}_car_k{p}{{}}_r{r}" for p in sl.paths for r in range(R)]         # branch (p, r) at group index p_local*R + r
        c1n, c2n = [b.format(1) for b in brs], [b.format(2) for b in brs]
        xd = dev(x, sl.cin_p)
        sl._scratch_pool, sl._unp_key, sl._unpack_merged = None, None, False
        s1, s2, tab1, tab2 = sl._unpack_tables(torch.device(DEV))

        # ---- the P*R 1x1 convs as ONE GEMM (:162)
        leaf = [fresh(P[n + ".kernel"], P[n + ".bias"]) for n in c1n]
        xl, = fresh(x)
        refs = [O.conv2d_same(xl, w, b) for w, b in leaf]
        u_raw = ops.conv2d_fwd(xd, sl.w1_f, sl.b1, 1, 1, ops.new_act(B, H, W, sl.Up, DEV))
        check(tag + "grouped conv1 fwd", u_raw[..., :sl.U], bf(torch.cat(refs, 3)), worst=worst)
        assert sl.Up == sl.U or u_raw[..., sl.U:].abs().max().item() == 0
        dys = [g_of(rec.conv[n][1]) for n in c1n]
        grads = torch.autograd.grad(refs, [xl] + [w for w, _ in leaf], dys)
        du_raw = cat_dev(dys, sl.Up)
        dxr = bf(torch.randn(B, H, W, sl.cin, dtype=torch.float64))                    # the running dx of the stage (shortcut / earlier slab)
        dx = ops.conv2d_dgrad(du_raw, sl.w1_d, 1, 1, ops.new_act(B, H, W, sl.cin_p, DEV), dev(dxr, sl.cin_p))
        check(tag + "grouped conv1 dgrad (+ running dx)", dx[..., :sl.cin], bf(grads[0] + dxr), worst=worst)
        net.flat.zero_grad()
        ops.fill_f32(s1, 0.0)
        ops.conv2d_wgrad(xd, du_raw, 1, 1, s1)
        ops.unpack_wgrad_batched(tab1)
        torch.cuda.synchronize()
        for gi, n in enumerate(c1n):
            check(f"{n} wgrad", getattr(m, n).kernel.grad, grads[1 + gi], worst=worst)

        # ---- BatchNorm + ELU over all branches in one launch (:164-165 / :169-170)
        def group_bn(names, width, gam, bet, mean, var, dgam, dbet, dbias, label, with_gap):
            xs = [rec.norm[n + "bn"][0].detach() for n in names]
            lv = [fresh(xq, P[n + "bn.gamma"], P[n + "bn.beta"]) for xq, n in zip(xs, names)]
            refs_ = [O.elu(bn_ref(xq, P, n + "bn", gq, bq)) for (xq, gq, bq), n in zip(lv, names)]
            C = sum(t.shape[-1] for t in xs)
            xin = cat_dev(xs, width)
            if with_gap:
                out, gap = ops.norm_act_fwd_gap(xin, C, gam, bet, torch.empty_like(xin), 1, 1, e, ops.ACT_ELU, 1.0, mean, var)
            else:
                out, gap = ops.norm_act_fwd(xin, C, gam, bet, torch.empty_like(xin), 1, 1, e, ops.ACT_ELU, 1.0, mean, var), None
            check(tag + label + " BN+ELU fwd", out[..., :C], bf(torch.cat(refs_, 3)), worst=worst)
            assert width == C or out[..., C:].abs().max().item() == 0
            if gap is not None:
                pooled = gap[0].sum(dim=1)[:, :C] / (H * W)
                check(tag + label + " pooled partial rows", pooled, out[..., :C].double().mean(dim=(1, 2)), worst=worst)
            dys_ = [g_of(rec.act_after(rec.norm[n + "bn"][1])) for n in names]
            gr = torch.autograd.grad(refs_, [t for l in lv for t in l], dys_)
            net.flat.zero_grad()
            dxn = ops.norm_act_bwd(xin, cat_dev(dys_, width), C, gam, bet, torch.empty_like(xin), dgam, dbet, 1, 1, e, ops.ACT_ELU, 1.0, mean, var,
                                   dbias=dbias)
            torch.cuda.synchronize()
            check(tag + label + " BN+ELU dx", dxn[..., :C], bf(torch.cat(gr[0::3], 3)), worst=worst)
            check(tag + label + " dgamma", dgam[:C], torch.cat(gr[1::3]), worst=worst)
            check(tag + label + " dbeta", dbet[:C], torch.cat(gr[2::3]), worst=worst)
            check(tag + label + " conv bias grad", dbias[:C], torch.cat([g_.sum(dim=(0, 1, 2)) for g_ in gr[0::3]]), worst=worst)
            return out, gap
        group_bn(c1n, sl.Up, sl.g1, sl.be1, sl.m1, sl.v1, sl.dg1, sl.dbe1, sl.db1, "conv1", False)

        # ---- the P*R 3x3 convs as ONE block-diagonal implicit GEMM (:167)
        us = [rec.conv[n][0].detach() for n in c2n]
        leaf2 = [fresh(P[n + ".kernel"], P[n + ".bias"]) for n in c2n]
        uls = fresh(*us)
        refs = [O.conv2d_same(ul, w, b) for ul, (w, b) in zip(uls, leaf2)]
        ud = cat_dev(us, sl.Up)
        v_raw = ops.conv2d_fwd(ud, sl.w2_f, sl.b2, sl.k, 1, ops.new_act(B, H, W, sl.Vp, DEV))
        check(tag + "grouped conv2 fwd", v_raw[..., :sl.V], bf(torch.cat(refs, 3)), worst=worst)
        assert sl.Vp == sl.V or v_raw[..., sl.V:].abs().max().item() == 0
        dys = [g_of(rec.conv[n][1]) for n in c2n]
        grads = torch.autograd.grad(refs, uls + [w for w, _ in leaf2], dys)
        dv = cat_dev(dys, sl.Vp)
        du = ops.conv2d_dgrad(dv, sl.w2_d, sl.k, 1, torch.empty_like(ud))
        check(tag + "grouped conv2 dgrad", du[..., :sl.U], bf(torch.cat(grads[:len(us)], 3)), worst=worst)
        net.flat.zero_grad()
        ops.fill_f32(s2, 0.0)
        ops.conv2d_wgrad(ud, dv, sl.k, 1, s2)
        ops.unpack_wgrad_batched(tab2)
        torch.cuda.synchronize()
        for gi, n in enumerate(c2n):
            check(f"{n} wgrad", getattr(m, n).kernel.grad, grads[len(us) + gi], worst=worst)
        y_gpu, gap = group_bn(c2n, sl.Vp, sl.g2, sl.be2, sl.m2, sl.v2, sl.dg2, sl.dbe2, sl.db2, "conv2", True)

        # ---- split attention, R distinct branches, BatchNorm-mode MLP, one dense2 per radix (:175-207).  Run as the model runs it:
        # on the y AND the pooled partial rows the norm launch just produced (so the fp64 expectation is evaluated on that y)
        ys = [y_gpu[..., gi * sl.cvkk:(gi + 1) * sl.cvkk].double().cpu() for gi in range(sl.G)]
        yls = fresh(*ys)
        Pl, mlp_leaves = dict(P), []
        for p in sl.paths:
            pre = f"{name}_car_k{p}_att"
            nm = [pre + "1.kernel", pre + "1.bias", pre + "_bn.gamma", pre + "_bn.beta"] + [f"{pre}2_r{r}.{w}" for r in range(R) for w in ("kernel", "bias")]
            lv = fresh(*[P[n] for n in nm])
            Pl.update(dict(zip(nm, lv)))
            mlp_leaves.append((nm, lv))
        refs = [O.archA_split_attention(yls[pi * R:(pi + 1) * R], Pl, f"{name}_car_k{p}_att") for pi, p in enumerate(sl.paths)]
        c1 = ops.new_act(B, H, W, c1_w, DEV, zero=True)
        o = sl.paths[0] * st.cvkk
        out = c1[..., o:o + sl.Co]
        d = sl._sa_desc(B, H * W, out)
        _, g, s, ws = ops.splitattn_fwd(d, y_gpu, sl.mlp_p, out, gap=gap)
        check(tag + "split attention fwd", out, bf(torch.cat(refs, 3)), worst=worst)
        out2 = ops.new_act(B, H, W, sl.Co, DEV)
        ops.splitattn_fwd(sl._sa_desc(B, H * W, out2), y_gpu, sl.mlp_p, out2)          # the pooling-pass form (no partial rows) agrees
        check(tag + "split attention fwd (own pooling pass)", out2, bf(torch.cat(refs, 3)), worst=worst)
        douts = [g_of(rec.sa[f"{name}_car_k{p}_att"][1]) for p in sl.paths]
        grads = torch.autograd.grad(refs, yls + [t for _, lv in mlp_leaves for t in lv], douts)
        d_c1 = ops.new_act(B, H, W, c1_w, DEV, zero=True)
        d_c1[..., o:o + sl.Co] = cat_dev(douts)
        dsl = d_c1[..., o:o + sl.Co]
        net.flat.zero_grad()
        dyd = ops.splitattn_bwd(sl._sa_desc(B, H * W, dsl), y_gpu, dsl, sl.mlp_p, sl.mlp_g, g, s, ws, torch.empty_like(y_gpu))
        torch.cuda.synchronize()
        check(tag + "split attention dy", dyd[..., :sl.V], bf(torch.cat(grads[:sl.G], 3)), 2 * TOL, worst=worst)   # dy = dout*s_r + broadcast(pool path): two roundings meet
        k0 = sl.G
        for nm, lv in mlp_leaves:
            for n, want in zip(nm, grads[k0:k0 + len(nm)]):
                mod, attr = n.rsplit(".", 1)
                got = getattr(getattr(m, mod), attr).grad
                check(f"split attention d {n}", got.reshape(want.shape), want, 2 * TOL, worst=worst)
            k0 += len(nm)

    # ---- shortcut conv + BN + ELU where channel counts differ (:142-145), concats_2 with the residual in its epilogue (:140,:148)
    if st.has_sc:
        conv_check(net, rec, P, name + "_cc", st.cc, worst, skip_bias=True)
        nb = name + "_scbn"
        xs = rec.norm[nb][0].detach()
        xl, gl, bl = fresh(xs, P[nb + ".gamma"], P[nb + ".beta"])
        ref = O.elu(bn_ref(xl, P, nb, gl, bl))
        sc = st.scbn.forward(dev(xs), ops.ACT_ELU, 1.0)
        check(nb + " BN+ELU fwd", sc, bf(ref), worst=worst)
        dy = g_of(rec.act_after(rec.norm[nb][1]))
        gx, gg, gb = torch.autograd.grad(ref, [xl, gl, bl], dy)
        net.flat.zero_grad()
        dsc = st.scbn.backward(dev(dy), dbias=st.cc.bias.grad)
        torch.cuda.synchronize()
        check(nb + " dx", dsc, bf(gx), worst=worst)
        check(nb + " dgamma", st.scbn.gamma.grad, gg, worst=worst)
        check(nb + " dbeta", st.scbn.beta.grad, gb, worst=worst)
        check(name + "_cc dbias (sum dx)", st.cc.bias.grad, gx.sum(dim=(0, 1, 2)), worst=worst)
        res = rec.act_after(rec.norm[nb][1]).detach()
    else:
        res = x
    conv_check(net, rec, P, name + "_concats_2", st.concats_2, worst, residual=res)
    print(f"Arch A {name} worst:", [(f"{e:.2e}", n) for e, n in sorted(worst, reverse=True)[:5]])


def test_decoder_head_and_loss_layer_by_layer(world):
    from ultrasound_modeling_amd import ops
    net, P, rec = world
    m = net.resModel
    worst = []
    for i, (name, oc, drop) in enumerate(m.UPS):                                                    # :109-122, :209-220
        layer, bn = getattr(m, name + "_t_conv"), getattr(m, name + "_bn")
        x, y_up, keep = rec.up[name]
        # 4x4 stride-2 transposed conv (:210): forward, backward-data, weight gradient straight into the Keras [k,k,Cout,Cin] variable
        tn = name + "_t_conv"
        xin, y_t = rec.conv[tn]
        xin = xin.detach()
        xl, wl, bl = fresh(xin, P[tn + ".kernel"], P[tn + ".bias"])
        ref_t = O.conv2d_transpose_s2_same(xl, wl, bl)
        raw = layer.forward(dev(xin))
        check(tn + " fwd", raw[..., :oc], bf(ref_t), worst=worst)
        dy_t = g_of(y_t)
        gx, gw, _ = torch.autograd.grad(ref_t, [xl, wl, bl], dy_t)
        net.flat.zero_grad()
        dx = m._tconv_backward(layer, dev(dy_t))
        torch.cuda.synchronize()
        check(tn + " dgrad", dx[..., :layer.cin], bf(gx), worst=worst)
        check(tn + " wgrad", layer.kernel.grad, gw, worst=worst)
        # BN + always-on dropout + ReLU written into the concat buffer's first `oc` channels (:213-218)
        nb = name + "_bn"
        xs = rec.norm[nb][0].detach()
        xl, gl, bl = fresh(xs, P[nb + ".gamma"], P[nb + ".beta"])
        pre = bn_ref(xl, P, nb, gl, bl)
        ref = torch.relu(pre * keep * 2.0 if keep is not None else pre)
        B, H2, W2, _ = xs.shape
        cat = ops.new_act(B, H2, W2, oc + 8, DEV, zero=True)
        mask = (keep * 2.0).to(torch.bfloat16).to(DEV) if keep is not None else None
        rawd = dev(xs)
        ops.norm_act_fwd(rawd, oc, bn.gamma.data, bn.beta.data, cat[..., :oc], 1, 1, 1e-3, ops.ACT_RELU, 0.0, bn.moving_mean_p, bn.moving_variance_p,
                         mask=mask)
        check(nb + (" +dropout" if drop else "") + " +ReLU fwd", cat[..., :oc], bf(ref), worst=worst)
        assert cat[..., oc:].abs().max().item() == 0
        dy = g_of(y_up)
        gx, gg, gb = torch.autograd.grad(ref, [xl, gl, bl], dy)
        dcat = ops.new_act(B, H2, W2, oc + 8, DEV, zero=True)
        dcat[..., :oc] = dev(dy)
        net.flat.zero_grad()
        draw = ops.norm_act_bwd(rawd, dcat[..., :oc], oc, bn.gamma.data, bn.beta.data, torch.empty_like(rawd), bn.gamma.grad, bn.beta.grad, 1, 1, 1e-3,
                                ops.ACT_RELU, 0.0, bn.moving_mean_p, bn.moving_variance_p, dbias=layer.bias.grad, mask=mask)
        torch.cuda.synchronize()
        check(nb + " dx", draw, bf(gx), worst=worst)
        check(nb + " dgamma", bn.gamma.grad, gg, worst=worst)
        check(nb + " dbeta", bn.beta.grad, gb, worst=worst)
        check(tn + " dbias (sum dx)", layer.bias.grad, gx.sum(dim=(0, 1, 2)), worst=worst)
    # ---- f_tran (4x4 stride-2, 160 -> 3) in quad form + softmax + my_loss_cat (:124-125, :234-248), and back
    name = "f_tran"
    x, _ = rec.conv[name]
    x = x.detach()
    xl, wl, bl = fresh(x, P[name + ".kernel"], P[name + ".bias"])
    logits_ref = O.conv2d_transpose_s2_same(xl, wl, bl)
    B, h, w, _ = x.shape
    logits = m._quad.forward(dev(x))                                                                 # fp32 [B,h,w,16]: slot 4*(2a+b)+class
    full = logits.view(B, h, w, 2, 2, 4).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, 4)
    check("f_tran logits", full[..., :3], logits_ref, worst=worst)
    lg = full[..., :3].double().cpu().requires_grad_(True)                                           # the loss kernel on the product's own fp32 logits
    lm_ref = O.my_loss_cat(rec.y, O.softmax_lastaxis(lg), 2 * h, 2 * w)
    lm_ref.sum().backward()                                                                          # tape.gradient of a map = gradient of its sum
    yd = rec.y.float().to(DEV).contiguous()
    probs = torch.empty((B, 2 * h, 2 * w, 3), dtype=torch.float32, device=DEV)
    scale = torch.zeros(4 * h * w * 3, dtype=torch.float32, device=DEV)
    lmap = torch.zeros(4 * h * w, dtype=torch.float32, device=DEV)
    dl4 = ops.new_act(B, h, w, 16, DEV, zero=True)
    ops.loss_cat_scale(yd, scale)
    ops.softmax_loss(logits, yd, probs, lmap, dl4, HW=4 * h * w, C_classes=3, loss_kind=1, scale=scale, quad_w=2 * w)
    check("softmax probs", probs, O.softmax_lastaxis(lg.detach()), worst=worst)
    check("my_loss_cat map", lmap.reshape(2 * h, 2 * w), lm_ref, worst=worst)
    dfull = dl4.view(B, h, w, 2, 2, 4).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, 4)
    check("softmax + my_loss_cat dlogits", dfull[..., :3], bf(lg.grad), worst=worst)
    dy = bf(dfull[..., :3].double().cpu())
    gx, gw, gb = torch.autograd.grad(logits_ref, [xl, wl, bl], dy)
    net.flat.zero_grad()
    with ops.overlap_region():
        dx = m._quad.backward(dl4)
    torch.cuda.synchronize()
    check("f_tran dgrad", dx[..., :x.shape[-1]], bf(gx), worst=worst)
    check("f_tran wgrad", m.f_tran.kernel.grad, gw, worst=worst)
    check("f_tran dbias", m.f_tran.bias.grad, gb, worst=worst)
    print("Arch A decoder worst:", [(f"{e:.2e}", n) for e, n in sorted(worst, reverse=True)[:6]])
