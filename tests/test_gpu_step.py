"""GPU tests of the step paths that are TIMED and SHIPPED (not only the eager single-GPU step):

  * HIP-graph replay == eager (bench.py times the replay): Arch B and Arch A, three optimisation steps on varying batches -
    loss, parameters, Adam moments and the device step counter agree; Arch A's always-on dropout draws the same mask for the
    same optimisation step on both paths and fresh masks from replay to replay;
  * ``capture_graph`` does not consume optimisation steps;
  * the data-parallel path on ONE GPU through a real RCCL process group of world size 1 (MainParallel.MirroredTrainer ->
    flat.AdamClip.clip_local -> all-reduce -> AdamClip.apply(already_clipped)), eager, as two HIP graphs around the
    collective, and with the chunk-pipelined exchange, against the plain single-replica step at a gradient norm > 1 so the clip
    is active (VisionTransformer.py:244-245; MainParallel.py:117-146);
  * the public loss methods ``compute_loss`` / ``my_loss_cat`` (VisionTransformer.py:225-227; TBI_ResNest.py:234-248);
  * save / load round trip including the optimiser state.

Since round 2 no kernel of the step uses float atomics at these sizes (ordered grid sums for the loss and the squared
gradient norm, slice-parallel split-K finishing), so the step is bitwise reproducible and "equal" below means EQUAL BITS:
eager == eager on a second model == HIP-graph replay == the data-parallel forms (whose `g*clip` happens in a separate kernel
but is the same single fp32 multiplication).
"""
import os

import pytest
import torch

import usseg_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batches(n, B, H, W):
    return [tuple(t for t in O.synthetic_batch(B, H, W, 1, seed=40 + i)) for i in range(n)]


def _arch_b(seed=3):
    from ultrasound_modeling_amd.VisionTransformer import VisionTransformer
    P = {k: v.float().double() for k, v in O.init_vision_transformer_params(channel=1, seed=seed, perturb=True).items()}
    net = VisionTransformer(batch_size=2, img_size=(64, 64), in_channels=1, learning_rate=1e-3)
    net.load_params(P)
    return net


def _arch_a(seed=4):
    from ultrasound_modeling_amd.TBI_ResNest import ResNest
    P = {k: v.float().double() for k, v in O.init_archA_params(channel=1, radix=3, kpaths=4, seed=seed, perturb=True).items()}
    net = ResNest(64, 64, 1, 3, ksize=3, radix=3, kpaths=4, learning_rate=5e-3)
    net.load_params(P)
    return net


def _state(net):
    torch.cuda.synchronize()
    return dict(p=net.flat.flat.clone(), m=net.optimizer.m.clone(), v=net.optimizer.v.clone(), step=int(net.optimizer.step_dev.item()))


def _assert_same_state(a, b, what):
    assert a["step"] == b["step"], what
    for k in ("p", "m", "v"):
        if not torch.equal(a[k], b[k]):
            d = (a[k].double() - b[k].double()).abs()
            raise AssertionError(f"{what}: {k} differs in {(d > 0).sum().item()} of {d.numel()} entries (max abs {d.max().item():.3e})")


def test_graph_replay_equals_eager_arch_b():
    eager, graph = _arch_b(), _arch_b()
    bs = _batches(3, 2, 64, 64)
    graph.capture_graph(bs[0][0], bs[0][1].float())
    _assert_same_state(_state(eager), _state(graph), "capture must not change the training state")
    for i, (x, y) in enumerate(bs):
        l0, p0 = eager.train_step(x, y.float())
        l1, p1 = graph.train_step(x, y.float())
        torch.cuda.synchronize()
        assert l0.item() == l1.item(), (i, l0.item(), l1.item())
        assert torch.equal(p0, p1), i
        _assert_same_state(_state(eager), _state(graph), f"Arch B step {i}")
    assert _state(graph)["step"] == 3


def test_graph_replay_equals_eager_arch_a_with_random_dropout():
    eager, graph = _arch_a(), _arch_a()
    assert eager.resModel.injected_masks is None            # the raw tf.nn.dropout(0.5) of TBI_ResNest.py:216, not injected
    bs = _batches(3, 2, 64, 64)
    graph.capture_graph(bs[0][0], bs[0][1].float())
    _assert_same_state(_state(eager), _state(graph), "capture must not change the training state")
    masks = []
    for i, (x, y) in enumerate(bs):
        lm0, a0, p0 = eager.step(x, y.float(), train=True)
        m_e = [m.clone() for m in eager.resModel._masks if m is not None]
        lm1, a1, p1 = graph.step(x, y.float(), train=True)
        torch.cuda.synchronize()
        m_g = [m.clone() for m in graph.resModel._masks if m is not None]
        assert len(m_e) == 3 and all(torch.equal(a, b) for a, b in zip(m_e, m_g)), f"step {i}: eager and replayed masks differ"
        masks.append(m_g)
        assert torch.equal(lm0, lm1) and torch.equal(p0, p1), i
        _assert_same_state(_state(eager), _state(graph), f"Arch A step {i}")
    for a, b in zip(masks[0], masks[1]):                    # a replay draws a FRESH mask (device step counter in the seed)
        assert ((a > 0) != (b > 0)).float().mean().item() > 0.4


# ------------------------------------------------------------------------------------------------ data parallel, one GPU
@pytest.fixture(scope="module")
def rccl_world1():
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        port = 29600 + os.getpid() % 300
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)   # "nccl" is RCCL on ROCm
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["eager", "two_graphs", "chunked_eager", "chunked_graph"])
def test_data_parallel_path_on_one_gpu(rccl_world1, mode):
    """MirroredTrainer over an in-process RCCL group of one rank: per-replica clip -> all-reduce -> already-clipped Adam must
    equal the plain step (clip inside the Adam kernel) when the clip is ACTIVE."""
    from ultrasound_modeling_amd.MainParallel import MirroredTrainer
    plain, dp = _arch_b(seed=5), _arch_b(seed=5)
    tr = MirroredTrainer(dp, force=True, chunks=3 if mode.startswith("chunked") else 1)
    assert dp.grad_sync is not None and (getattr(dp.grad_sync, "chunks", None) is not None) == mode.startswith("chunked")
    bs = _batches(3, 2, 64, 64)
    if mode.endswith("graph") or mode == "two_graphs":
        dp.capture_graph(bs[0][0], bs[0][1].float())
        g1, g2 = dp._graph
        assert (g2 is not None) == (mode == "two_graphs")
    for i, (x, y) in enumerate(bs):
        l0, p0 = plain.train_step(x, y.float())
        gn = float(plain.optimizer.sumsq[0].item()) ** 0.5
        assert gn > 1.0, "the test needs an active clip"
        l1, p1 = tr.train_step(x, y.float())
        torch.cuda.synchronize()
        assert l0.item() == l1.item()
        # data-parallel form: the flat gradient buffer holds the CLIPPED gradient (norm 1), the plain form the raw one
        gdp = float(dp.flat.grad.double().norm().item())
        assert abs(gdp - 1.0) < 1e-3, gdp
        _assert_same_state(_state(plain), _state(dp), f"{mode} step {i}")


def test_chunk_ranges_cover_the_buffer():
    from ultrasound_modeling_amd.step import even_chunks
    for n, k in ((6270904, 4), (1000, 4), (65536 * 3 + 5, 3), (7, 1)):
        r = even_chunks(n, k, 1 << 16)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r[:-1], r[1:])) and len(r) <= k
        assert all(lo % (1 << 16) == 0 for lo, _ in r)


# ------------------------------------------------------------------------------------------------ public loss methods
def test_compute_loss_on_probabilities():
    net = _arch_b()
    net.batch_size = 4                                       # the GLOBAL batch divides the loss (:227)
    g = torch.Generator().manual_seed(1)
    probs = torch.softmax(3 * torch.randn(2, 64, 64, 3, generator=g, dtype=torch.float64), -1)
    probs[0, 0, 0] = torch.tensor([1.0, 0.0, 0.0])           # exercises the 1e-7 clip
    probs = probs * (0.5 + torch.rand(2, 64, 64, 1, generator=g, dtype=torch.float64))    # not normalised: Keras divides by the class sum
    _, y = O.synthetic_batch(2, 64, 64, 1, seed=2)
    want = O.compute_loss(y, probs, 4)
    got = net.compute_loss(y.float(), probs.float())
    assert got.shape == () and abs(got.item() - want.item()) <= 2e-5 * abs(want.item()), (got.item(), want.item())


def test_my_loss_cat_on_probabilities():
    net = _arch_a()
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(3 * torch.randn(2, 64, 64, 3, generator=g, dtype=torch.float64), -1)
    _, y = O.synthetic_batch(2, 64, 64, 1, seed=4)
    want = O.my_loss_cat(y, probs, 64, 64)
    got = net.my_loss_cat(y.float(), probs.float())
    assert tuple(got.shape) == (64, 64)
    err = ((got.double().cpu() - want).norm() / want.norm()).item()
    assert err < 1e-5, err


# ------------------------------------------------------------------------------------------------ persistence
def test_save_load_round_trip_with_optimizer_state(tmp_path):
    a = _arch_b()
    x, y = _batches(1, 2, 64, 64)[0]
    a.train_step(x, y.float())
    a.save(str(tmp_path / "b.pt"))
    b = _arch_b(seed=9)
    b.load(str(tmp_path / "b.pt"))
    _assert_same_state(_state(a), _state(b), "Arch B save/load")
    la, _ = a.train_step(x, y.float())
    lb, _ = b.train_step(x, y.float())
    _assert_same_state(_state(a), _state(b), "Arch B step after load")
    n = _arch_a()
    n.step(x, y.float(), train=True)
    path = n.resModel.save(str(tmp_path / "a.pt"))          # the driver's call (TBI_ResNest.py:472)
    m = _arch_a(seed=11)
    m.load_params_file(path)
    _assert_same_state(_state(n), _state(m), "Arch A save/load")
