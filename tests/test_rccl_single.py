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
