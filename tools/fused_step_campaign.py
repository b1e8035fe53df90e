"""Offline campaign of tests/test_train_step.py::test_fused_step_equals_composed_step_on_random_configurations over many more seeds than the
suite runs (python tools/fused_step_campaign.py FIRST COUNT). Prints every failing seed with the first line of its assertion."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_train_step as T  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
# optional third argument: "control" = the composed step against ITSELF (what run-to-run noise alone does to the same bound), "fused2" = fused against fused
mode = sys.argv[3] if len(sys.argv) > 3 else "fused-vs-composed"
forms = {"control": (False, False), "fused2": (True, True)}.get(mode, (False, True))
fails = []
for seed in range(first, first + count):
    try:
        T.test_fused_step_equals_composed_step_on_random_configurations(seed, forms, strict=True)
    except Exception as e:  # noqa: BLE001
        fails.append((seed, type(e).__name__, str(e).splitlines()[0][:300] if str(e) else ""))
        print("FAIL", seed, type(e).__name__, (str(e).splitlines()[0][:300] if str(e) else ""), flush=True)
        if not isinstance(e, AssertionError):
            traceback.print_exc()
print(f"{mode} campaign, seeds {first}-{first + count - 1}: {count - len(fails)} of {count} cases equal; failures: {len(fails)}")
for f in fails:
    print("   ", f)
