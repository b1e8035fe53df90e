"""VERDICT r4 #4 / SURVEY.md 8(e): the `nccl` (= RCCL on ROCm) backend executed on the GPU box — a world-size-1 process group in a
fresh child process (tests/helpers/rccl_world1.py): render_sharded through all_gather_into_tensor (bitwise == the direct render),
bench.py's asynchronous tile all-gather + wait, and allreduce_gradients(field=..., overlap=True) through the side-stream branch of
parallel.allreduce_buckets against the flat message. The child is killed by PID when the deadline passes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world_size_1_paths(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4",
               PYTHONFAULTHANDLER="1")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_world1.py")], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        p.kill()
        out, _ = p.communicate()
        pytest.fail("rccl_world1.py did not finish in 240 s:\n" + out[-4000:])
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in out, out[-4000:]
    assert "render_sharded == direct render" in out and "async all_gather_into_tensor" in out and "bitwise" in out


@pytest.mark.parametrize("mode", ["weak", "c4"])
def test_bench_py_over_rccl_with_one_rank(tmp_path, mode):
    """bench.py's own N > 1 code — RCCL group with device_id, object gather of the rank evidence, the tile all-gather inside the timed
    region (asynchronous in weak mode, inside every step + bitwise check in c4 mode), the max-over-ranks all-reduce, the data-parallel
    fused train step — with ONE rank on the one GPU (T2N_BENCH_FORCE_GROUP=1, backend nccl = RCCL): what the driver's SCALE run executes."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), T2N_BENCH_FORCE_GROUP="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8", T2N_BENCH_DEADLINE_S="200", PYTHONFAULTHANDLER="1")
    args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--train-iters", "2", "--train-warmup", "1"]
    if mode == "c4":
        args += ["--mode", "c4", "--check-c4", "--no-train"]
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        pytest.fail("bench.py over RCCL did not finish:\n" + err[-4000:])
    assert p.returncode == 0, err[-4000:]
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    r = d["config"]["rccl"]
    assert r["backend"] == "nccl" and r["is_rccl"] and r["world_size"] == 1 and d["config"]["rccl_version"].count(".") == 2
    assert "all-gather" in d["config"]["parallelism"] and d["value"] > 0
    if mode == "c4":
        assert d["scaling"] == "strong" and d["config"]["c4_gathered_equals_single_rank"] is True
    else:
        assert d["config"]["train_dp_iters_per_s"] > 0 and "all-reduce" in d["config"]["train_dp_step"]
