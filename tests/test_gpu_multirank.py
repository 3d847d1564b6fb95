"""The multi-rank bench path end to end on ONE GPU: `bench.py --gpus 2` launches its own two ranks (torch.distributed.run as a child), both
share device 0 and exchange through gloo (USSEG_DIST_BACKEND / USSEG_BENCH_SHARE_GPU: a rehearsal, never a measurement).  What it pins:
rank launch, rank-0 weight broadcast, batch split, the step as two HIP graphs with the exchange between them, per-replica clip, the loss
reduction, max-over-ranks timing and ONE JSON line from rank 0 (MainParallel.py:117-146,209-210 semantics; the RCCL transport itself is
covered at world size 1 in test_gpu_step.py and at world size 2 over gloo on the CPU in test_cpu_host.py)."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a HIP device")
def test_bench_two_ranks_on_one_gpu_over_gloo():
    env = dict(os.environ, USSEG_DIST_BACKEND="gloo", USSEG_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--profile-steps", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    assert d["scaling"] == "weak" and d["config"]["hip_graph"] is True
    assert math.isfinite(d["config"]["final_loss"]) and d["value"] > 0
