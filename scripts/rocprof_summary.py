#!/usr/bin/env python3
"""Turn a rocprofv3 results database (rocprofv3 7.x writes <name>_results.db) into the per-kernel
summary CSVs kept under profiles/: kernel-trace stats, and (if present) PMC counters per kernel."""
import collections
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
cur = db.cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out + "_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, f"{tot:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
print(f"wrote {out}_kernel_stats.csv ({len(rows)} kernels)")
n = cur.execute("select count(*) from counters_collection").fetchone()[0]
if n:
    disp = collections.OrderedDict()
    for d, kn, gs, cn, val, dur in cur.execute(
            "select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection"):
        e = disp.setdefault(d, {"k": kn, "grid": gs, "dur": dur, "c": collections.Counter()})
        e["c"][cn] += val
    agg = collections.OrderedDict()
    names = set()
    for e in disp.values():
        a = agg.setdefault((e["k"], e["grid"]), {"n": 0, "dur": 0.0, "c": collections.Counter()})
        a["n"] += 1
        a["dur"] += e["dur"]
        a["c"].update(e["c"])
        names.update(e["c"].keys())
    names = sorted(names)
    with open(out + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_size", "dispatches", "avg_us"] + names)
        for (k, g), a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
            w.writerow([k, g, a["n"], f"{a['dur'] / a['n'] / 1e3:.2f}"] + [f"{a['c'][c] / a['n']:.0f}" for c in names])
    print(f"wrote {out}_pmc.csv ({len(agg)} kernel/grid rows, counters: {names})")

# GPU idle between consecutive kernels (launch gaps): dispatch timeline from the `kernels` view.
try:
    tl = cur.execute("select start, end, name from kernels order by start").fetchall()
except sqlite3.Error as e:
    tl = []
    print("no kernel timeline:", e, [r[0] for r in cur.execute("select name from sqlite_master where type='view'")])
if tl:
    busy = sum(e - s0 for s0, e, _ in tl)
    gaps = [tl[i + 1][0] - tl[i][1] for i in range(len(tl) - 1)]
    small = [g for g in gaps if 0 <= g < 50_000]          # < 50 us: back-to-back launches inside a forward pass
    neg = sum(1 for g in gaps if g < 0)
    # true idle time: the span minus the UNION of the busy intervals (kernels of several streams overlap), split at pauses > 5 ms
    # (the host-side gaps between warm-up, timed loop and teardown are not launch gaps)
    idle, span, cur_end, seg_start = 0, 0, None, None
    for s0, e, _ in tl:
        if cur_end is None:
            cur_end, seg_start = e, s0
            continue
        if s0 > cur_end:
            if s0 - cur_end > 5_000_000:
                span += cur_end - seg_start
                seg_start = s0
            else:
                idle += s0 - cur_end
        cur_end = max(cur_end, e)
    span += cur_end - seg_start
    with open(out + "_gaps.txt", "w") as f:
        msg = (f"dispatches {len(tl)}  kernel-busy {busy / 1e6:.3f} ms  gaps<50us: n={len(small)} total {sum(small) / 1e6:.3f} ms "
               f"mean {sum(small) / max(1, len(small)) / 1e3:.2f} us  overlapping dispatches {neg}  "
               f"active span {span / 1e6:.3f} ms of which no kernel running {idle / 1e6:.3f} ms ({100.0 * idle / max(1, span):.1f} %)")
        f.write(msg + "\n")
        print(msg)
