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

# ---- per-queue view: which launches form the serial chain of the step, and how much of it is gaps ----
byq = defaultdict(list)
for n, s, e, g, w, q in win:
    byq[q].append((s, e, short(n), g // max(w, 1)))
print("\nper queue (hardware queue = one stream of the replayed graph):")
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    lst.sort()
    busy_q = sum(e - s for s, e, _, _ in lst)
    gaps = [max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)]
    small = [x for x in gaps if x < 50_000]  # gaps inside a step (the step boundary itself is longer)
    print(f"queue {q}: {len(lst)/steps:.0f} kernels/step, busy {busy_q/steps/1e6:.3f} ms/step, "
          f"gaps<50us {sum(small)/steps/1e6:.3f} ms/step (median {sorted(small)[len(small)//2]/1e3 if small else 0:.2f} us)")
for main_q in sorted(byq, key=lambda q: -len(byq[q])):
    tq = defaultdict(lambda: [0, 0.0, 0])
    for s, e, k, wg in byq[main_q]:
        tq[k][0] += 1; tq[k][1] += e - s; tq[k][2] += wg
    print(f"\nqueue {main_q} by kernel:")
    print(f"{'kernel':60s} {'n/step':>7s} {'ms/step':>8s} {'avg us':>8s} {'avg WGs':>8s}")
    for k, (c, t, wg) in sorted(tq.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{k:60s} {c/steps:7.1f} {t/steps/1e6:8.3f} {t/c/1e3:8.2f} {wg/c:8.0f}")
# how do the queues interleave in time?  (first/last launch of each queue inside the last step)
last = [r for r in win if r[1] >= marks[-2]]
for q in sorted(set(r[5] for r in last)):
    ss = [r for r in last if r[5] == q]
    print(f"queue {q}: first launch +{(ss[0][1]-marks[-2])/1e6:.3f} ms, last end +{(ss[-1][2]-marks[-2])/1e6:.3f} ms after the step's AdamW start")
