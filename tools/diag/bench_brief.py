"""Diagnostic: one line per bench.py JSON file (value, ms/step, single-stream ms/step, per-kernel launch us)."""
import json, sys
for f in sys.argv[1:]:
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    k = {n: v["avg_launch_us"] for n, v in (l.get("mfma_kernels") or {}).items()}
    t = l.get("train_step") or {}
    print(f"{f}: value {l['value']:.0f} ms/step {l['ms_per_step']} [{l['ms_per_step_min_max']}] single {l.get('ms_per_step_single_stream')} "
          f"train {t.get('ms_per_step')} kernels {k}")
