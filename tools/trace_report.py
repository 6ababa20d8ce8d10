"""Per-step report from a rocprofv3 rocpd database (kernel trace): picks the steady-state graph replays
(the last N periodic windows), prints busy time / union time / concurrency and the per-symbol totals."""
import re, sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, queue_id from kernels order by start").fetchall()
print("dispatches", len(rows))
# steady-state window: find the adamw_clip launches (one per step) and take the last `steps` periods
marks = [r[1] for r in rows if "adamw_clip" in r[0]]
print("adamw launches", len(marks))
lo, hi = marks[-steps - 1], marks[-1]
win = [r for r in rows if lo <= r[1] < hi]
span = (hi - lo) / steps
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]
tot = defaultdict(lambda: [0, 0.0])
for n, s, e, g, w, q in win:
    k = short(n); tot[k][0] += 1; tot[k][1] += e - s
busy = sum(v[1] for v in tot.values()) / steps
# union of intervals
ev = sorted((s, e) for _, s, e, _, _, _ in win)
u, cs, ce = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f"step period {span/1e6:.3f} ms; kernels/step {len(win)/steps:.0f}; sum of kernel time {busy/1e6:.3f} ms; union {u/steps/1e6:.3f} ms; "
      f"idle {(span-u/steps)/1e6:.3f} ms; queues {sorted(set(r[5] for r in win))}")
print(f"{'kernel':60s} {'n/step':>7s} {'ms/step':>8s} {'avg us':>8s}")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{k:60s} {c/steps:7.1f} {t/steps/1e6:8.3f} {t/c/1e3:8.2f}")
