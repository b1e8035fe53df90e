"""the memory the renderer and the train step take from the GPU is bounded. A fresh child process
(tests/helpers/memory_probe.py; torch.cuda.memory_reserved is process-wide) renders C2 frames and runs 300 fused train steps of the
bench's C3 loop, the run whose appearance list triples on the way; bounds: one C2 frame <= 6 GiB reserved, the training run <= 16 GiB."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reserved_memory_is_bounded():
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "memory_probe.py"), "300"], cwd=ROOT,
                         env=dict(os.environ, OMP_NUM_THREADS="8"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=400)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        pytest.fail("memory_probe.py did not finish:\n" + err[-3000:])
    assert p.returncode == 0, err[-3000:]
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(d)
    assert d["finite"] and d["frame_list_retry"] == 0
    assert d["frame_reserved_GiB"] <= 6.0, d
    assert d["train_peak_appearance_samples"] > 250000, d      # the run did go through its fog phase
    assert d["train_reserved_GiB_peak"] <= 16.0, d
