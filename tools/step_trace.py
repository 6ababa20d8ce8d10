"""Timeline of ONE replayed step from a rocprofv3 kernel trace (csv):  python tools/step_trace.py <kernel_trace.csv> [which]
Prints start offset (us), duration (us), gap to the previous kernel on the same queue, queue, workgroups, kernel --
steps are delimited by the `sumsq_kernel` launch that opens each deferred update (`which`: index of the delimiter; default: the
last step whose launch count is the most common one -- the tail of a bench run holds a shorter last step and the final flush).
NOTE: the profiler serialises side-stream branches more than the un-profiled replay does: read durations and
the order of the chain from this, not branch start times."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:56]


def main():
    path = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                         grid // max(wg, 1)))
    rows.sort()
    # a step opens with the sum of squares in front of the update (with the early partial sum of the GPS backbone's range --
    # engine.EARLY_SUMSQ -- there are more sumsq launches per step: the opening one is followed by the update kernel)
    def opens(i):
        if not rows[i][3].startswith("sumsq_kernel"):
            return False
        prev = [r[3] for r in rows[max(0, i - 200):i] if r[2] == rows[i][2]]
        if prev and prev[-1].startswith("sumsq_kernel"):
            return False
        nxt = [r[3] for r in rows[i + 1:i + 200] if r[2] == rows[i][2] and not r[3].startswith("sumsq_kernel")]  # same queue
        return bool(nxt) and nxt[0].startswith("adamw_clip")
    marks = [i for i in range(len(rows)) if opens(i)]
    if which is None:
        from collections import Counter
        lens = [b - a for a, b in zip(marks[:-1], marks[1:])]
        common = Counter(lens).most_common(1)[0][0]
        which = max(i for i, n in enumerate(lens) if n == common) - len(marks)
    lo, hi = marks[which], marks[which + 1] if which + 1 < 0 or which + 1 < len(marks) else len(rows)
    step = rows[lo:hi]
    t0 = step[0][0]
    last_end = defaultdict(lambda: None)
    print(f"# step of {len(step)} kernels, {(rows[hi][0] - t0) / 1e3:.1f} us from its first launch to the next step's")
    busy = defaultdict(float)
    for s, e, q, k, wgs in step:
        gap = (s - last_end[q]) / 1e3 if last_end[q] is not None else 0.0
        last_end[q] = e
        busy[q] += (e - s) / 1e3
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f} q{q} wg {wgs:5d} {k}")
    print("# busy us per queue:", {q: round(v, 1) for q, v in sorted(busy.items())})


if __name__ == "__main__":
    main()
