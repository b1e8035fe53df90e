"""bench.py's N > 1 code path (ray-tile render + asynchronous all-gather + data-parallel train step, and the strong-scaling
`--mode c4` frame) exercised with two ranks on ONE GPU over gloo (T2N_BENCH_BACKEND=gloo, T2N_BENCH_SAME_DEVICE=1): RCCL
cannot place two ranks on one device, and the real multi-GPU run belongs to the driver — this keeps that path from
regressing. The numbers are meaningless; the JSON contract and the collectives are what is checked.

Runs LAST (file name + tests/conftest.py ordering) and bounded: the two ranks are plain child processes with explicit
RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 (no elastic agent, no hostname lookups), every rank arms
faulthandler.dump_traceback_later (T2N_BENCH_DEADLINE_S) and writes its own log, and the test kills exactly the PIDs it
started when its own deadline passes — a hang costs minutes and prints where every rank stood."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline")


def run_ranks(tmp_path, bench_args, world=2, deadline_s=150.0, rank_deadline_s=120.0):
    """Start `world` ranks of bench.py as direct children; returns (returncodes, stdouts, stderrs). Never raises on a hang:
    the ranks are killed by PID and their logs come back for the assertion message."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), T2N_BENCH_BACKEND="gloo", T2N_BENCH_SAME_DEVICE="1", OMP_NUM_THREADS="4",
                   GLOO_SOCKET_IFNAME="lo", T2N_BENCH_DEADLINE_S=str(rank_deadline_s), PYTHONFAULTHANDLER="1")
        fo, fe = open(tmp_path / f"rank{r}.out", "w+"), open(tmp_path / f"rank{r}.err", "w+")
        files.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + bench_args,
                                      env=env, cwd=ROOT, stdout=fo, stderr=fe, start_new_session=True))
    t_end = time.time() + deadline_s
    while time.time() < t_end and any(p.poll() is None for p in procs):
        time.sleep(0.5)
    for p in procs:
        if p.poll() is None:
            p.kill()               # the exact child, nothing matched by pattern
            p.wait()
    outs, errs = [], []
    for fo, fe in files:
        fo.seek(0), fe.seek(0)
        outs.append(fo.read()), errs.append(fe.read())
        fo.close(), fe.close()
    return [p.returncode for p in procs], outs, errs


def report(rcs, outs, errs):
    return "\n".join(f"--- rank {r}: rc {rc}\nstdout: {o[-1500:]}\nstderr: {e[-3000:]}" for r, (rc, o, e) in
                     enumerate(zip(rcs, outs, errs)))


def check_finished(rcs, outs, errs):
    """rc 0 on every rank, or: a rank that never got as far as "gpu and process group ready" (two processes bringing up ONE
    device — a configuration only this test uses) is an environment stall, reported as a skip with the ranks' logs; anything
    after that marker (collectives, kernels, the JSON contract) is a failure."""
    if all(rc == 0 for rc in rcs):
        return
    if not all("gpu and process group ready" in e for e in errs):
        pytest.skip("two ranks could not bring up the one GPU of this box within the deadline (not the N>1 code path):\n"
                    + report(rcs, outs, errs))
    pytest.fail(report(rcs, outs, errs))


def one_json_line(outs):
    lines = [l for o in outs for l in o.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines            # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_over_gloo(tmp_path):
    rcs, outs, errs = run_ranks(tmp_path, ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--train-iters", "2",
                                           "--train-warmup", "1"])
    check_finished(rcs, outs, errs)
    d = one_json_line(outs)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and "workload" in d["config"] and "all-gather" in d["config"]["parallelism"]
    assert "train_dp_error" not in d["config"], d["config"].get("train_dp_error")
    assert d["config"]["train_dp_iters_per_s"] > 0 and "all-reduce" in d["config"]["train_dp_step"]
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
    ev = d["config"]["rccl"]                              # which devices the ranks ran on (here: gloo, both on device 0)
    assert ev["backend"] == "gloo" and ev["world_size"] == 2 and len(ev["ranks"]) == 2 and ev["distinct_devices"] == 1
    assert ev["all_gather_bytes_per_rank_per_step"] == 640000 * 16 and ev["all_reduce_bytes_per_train_step"] > 6e7


def test_c4_mode_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """BASELINE configs[3]: ONE 1600x1600 frame split into ray tiles, all-gather inside the timed region, gathered frame
    bitwise equal to the single-rank render (checked by bench.py itself on rank 0 with --check-c4)."""
    rcs, outs, errs = run_ranks(tmp_path, ["--mode", "c4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-train",
                                           "--check-c4"])
    check_finished(rcs, outs, errs)
    d = one_json_line(outs)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["c4_gathered_equals_single_rank"] is True
    assert "1600x1600" in d["config"]["workload"] and "interleaved 8-row bands" in d["config"]["workload_detail"]


def test_c4_mode_contiguous_tiles_two_ranks(tmp_path):
    rcs, outs, errs = run_ranks(tmp_path, ["--mode", "c4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-train",
                                           "--check-c4", "--c4-tiles", "contiguous"])
    check_finished(rcs, outs, errs)
    d = one_json_line(outs)
    assert d["config"]["c4_gathered_equals_single_rank"] is True and "contiguous row blocks" in d["config"]["workload_detail"]


def test_plain_bench_gpus_2_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two fresh ranks itself (before touching
    the GPU), relays rank 0's ONE JSON line and exits 0. Same-device / gloo here (a 1-GPU box); on an 8-GPU node the same command
    without the two T2N_BENCH_* variables runs one rank per GPU over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(T2N_BENCH_BACKEND="gloo", T2N_BENCH_SAME_DEVICE="1", GLOO_SOCKET_IFNAME="lo", T2N_BENCH_DEADLINE_S="150",
               T2N_BENCH_LAUNCH_DEADLINE_S="170", PYTHONFAULTHANDLER="1")
    with open(tmp_path / "err.log", "w+") as fe:
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                              "--no-cpu-baseline", "--no-train"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=fe, text=True,
                             start_new_session=True)
        try:
            out, _ = p.communicate(timeout=200)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        fe.seek(0)
        err = fe.read()
    if p.returncode != 0 and "gpu and process group ready" not in err:
        pytest.skip("two ranks could not bring up the one GPU of this box (not the launcher):\n" + err[-3000:])
    assert p.returncode == 0, err[-3000:]
    d = one_json_line([out])
    assert d["n_gpus"] == 2 and d["config"]["rccl"]["launcher"] == "bench.py self-launch" and len(d["config"]["rccl"]["ranks"]) == 2
