#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace): per kernel count / total / mean / max duration, the span of the
trace, and -- with --timeline N -- the first N dispatches after offset T (ms) as a timeline (start, duration, stream, name).
    python tools/rocpd_summary.py results.db [--timeline N [T_ms]]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, d.start, d.end, d.stream_id, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                  "on d.kernel_id = s.id order by d.start").fetchall()
if not rows:
    sys.exit("no kernel dispatches")
t0 = rows[0][1]
span = (max(r[2] for r in rows) - t0) / 1e6
agg = {}
for name, a, b, st, q in rows:
    short = name.split("(")[0].replace("ngsld::", "").replace("(anonymous namespace)::", "")[:70]
    e = agg.setdefault(short, [0, 0.0, 0.0])
    e[0] += 1
    e[1] += (b - a) / 1e6
    e[2] = max(e[2], (b - a) / 1e6)
print(f"{len(rows)} dispatches over {span:.1f} ms; sum of kernel durations {sum(v[1] for v in agg.values()):.1f} ms")
print(f"{'kernel':70s} {'calls':>6s} {'total ms':>10s} {'mean ms':>9s} {'max ms':>8s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:70s} {v[0]:6d} {v[1]:10.2f} {v[1] / v[0]:9.3f} {v[2]:8.3f}")
if "--timeline" in sys.argv:
    i = sys.argv.index("--timeline")
    n = int(sys.argv[i + 1])
    off = float(sys.argv[i + 2]) if len(sys.argv) > i + 2 else 0.0
    k = 0
    for name, a, b, st, q in rows:
        if (a - t0) / 1e6 < off:
            continue
        print(f"{(a - t0) / 1e6:10.3f} +{(b - a) / 1e6:8.3f} ms  stream {st} queue {q}  {name.split('(')[0][-60:]}")
        k += 1
        if k >= n:
            break
