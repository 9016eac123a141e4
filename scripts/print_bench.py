"""Print the headline numbers of a bench.py JSON line: `python scripts/print_bench.py FILE` or `... | python scripts/print_bench.py LABEL`."""
import json
import os
import sys

arg = sys.argv[1] if len(sys.argv) > 1 else "bench"
text = open(arg).read() if os.path.exists(arg) else sys.stdin.read()
d = json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
devs = d.get("config", {}).get("rank_devices") or []
print(arg, f"n_gpus={d.get('n_gpus')}", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms/step", f"ranks met: {len(devs)}")
tr = d.get("modes", {}).get("train")
if tr and "error" in tr:
    print("  train: ERROR", tr["error"])
elif tr:
    print("  train:", round(tr["images_per_s"], 1), "img/s", round(tr["ms_per_step"], 2), "ms/step", tr.get("gradient_exchange"),
          "exposed ms:", tr.get("exchange_exposed_ms_per_step"))
