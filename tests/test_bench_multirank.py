"""bench.py's N > 1 code path (ray-tile render + asynchronous all-gather + data-parallel train step) exercised with two ranks
on ONE GPU over gloo (T2N_BENCH_BACKEND=gloo, T2N_BENCH_SAME_DEVICE=1): RCCL cannot place two ranks on one device, and the
real multi-GPU run belongs to the driver — this keeps that path from regressing. The numbers are meaningless; the JSON
contract and the collectives are what is checked."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, T2N_BENCH_BACKEND="gloo", T2N_BENCH_SAME_DEVICE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and "workload" in d["config"] and "all-gather" in d["config"]["parallelism"]
    assert "train_dp_error" not in d["config"], d["config"].get("train_dp_error")
    assert d["config"]["train_dp_iters_per_s"] > 0 and "all-reduce" in d["config"]["train_dp_step"]
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
