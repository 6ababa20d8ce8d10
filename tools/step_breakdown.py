"""Steady-state kernel breakdown of the replayed train step from a rocprofv3 kernel trace (csv):
    python tools/step_breakdown.py <dir with *kernel_trace.csv> [n_steps]
Steps are delimited by the gradient-norm kernel (sumsq_kernel: one launch per step in every engine mode); the last n
complete steps are analysed: per-kernel time,
launches per step, busy time (union of kernel intervals) and idle gaps of the device."""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "sumsq_kernel" in r[2]]
if len(marks) < 3:
    marks = [i for i, r in enumerate(rows) if "adamw" in r[2].lower()]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3  # trailing marks to leave out: the final flush and the eager profile pass
assert len(marks) > n + 1 + skip, len(marks)
lo, hi = marks[-n - 1 - skip], marks[-1 - skip]  # [first launch of a step's head, first launch of the step n later)
win = rows[lo:hi]
t0, t1 = win[0][0], win[-1][1]
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"void ", "", k)
    return k.split("(")[0][:90]
tot, cnt = collections.Counter(), collections.Counter()
for s, e, k in win:
    tot[short(k)] += e - s
    cnt[short(k)] += 1
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = (t1 - t0) / n / 1e6
print(f"{n} steps: {wall:.3f} ms per step wall, device busy {busy / n / 1e6:.3f} ms, idle {(t1 - t0 - busy) / n / 1e6:.3f} ms, "
      f"sum of kernel durations {sum(tot.values()) / n / 1e6:.3f} ms, {len(win) / n:.0f} launches per step")
for k, v in tot.most_common(45):
    print(f"{v / n / 1e3:9.1f} us  {cnt[k] / n:7.1f} x {v / cnt[k] / 1e3:8.1f} us  {k}")

# exclusive time: how long is a kernel the ONLY one on the device (a proxy for the critical path of the replayed graph)
events = []
for i, (s, e, k) in enumerate(win):
    events.append((s, 1, i))
    events.append((e, 0, i))
events.sort()
active, last, excl = set(), None, collections.Counter()
for t, kind, i in events:
    if last is not None and len(active) == 1:
        excl[short(win[next(iter(active))][2])] += t - last
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
    last = t
print(f"\nexclusive time (only kernel on the device), total {sum(excl.values()) / n / 1e6:.3f} ms per step")
for k, v in excl.most_common(30):
    print(f"{v / n / 1e3:9.1f} us  {k}")
