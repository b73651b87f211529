"""Multi-GPU path on real devices: `bench.py --gpus 2` started WITHOUT a launcher must become two ranks over RCCL (nccl backend),
report n_gpus = 2 and verify its round trips.  Skips on a box with fewer than two GPUs (the round-end driver runs the 1/2/4/8
scaling bench on an 8-GPU node; this is the smoke test of the same command)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_nccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "8192",
                                   "--no-cpu-baseline", "--no-extra-configs"], env=env, timeout=900).decode()
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["verified"] is True and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "blocks sharded x2"
