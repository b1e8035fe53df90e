"""Offline fuzz campaign: the three randomised differential tests of tests/test_hip_fuzz.py over many more seeds than the suite
runs (python tools/fuzz_campaign.py FIRST COUNT). Prints every failing (test, seed) with the first line of its assertion."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_hip_fuzz as F  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for name in ("test_random_configuration_vs_c_oracle", "test_random_frames_on_the_tile_marcher_vs_c_oracle",
             "test_random_configuration_gradients_vs_oracle_autograd"):
    fn = getattr(F, name)
    for seed in range(first, first + count):
        try:
            fn(seed)
        except BaseException as e:  # noqa: BLE001
            msg = (str(e).strip().splitlines() or [repr(e)])[0][:300]
            bad.append((name, seed, type(e).__name__, msg))
            print("FAIL", name, seed, type(e).__name__, msg, flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
    print("done", name, flush=True)
print("failures:", len(bad))
for b in bad:
    print(b)
