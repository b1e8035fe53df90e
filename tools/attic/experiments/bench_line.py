"""One render-only bench run, printing ms/frame and the per-kernel split (for A/B runs under swap_run.sh)."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-train", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
print(round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["config"]["kernel_ms_per_frame"].items()})
