"""From a rocprofv3 rocpd database of bench.py: per kernel symbol, time spent in launches with few workgroups
(<= 260: latency-bound, one workgroup per CU at most) vs many.   python tools/small_grid_report.py <db> [steps]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
marks = [r[1] for r in rows if "adamw_clip" in r[0]]
lo, hi = marks[-steps - 1], marks[-1]
win = [r for r in rows if lo <= r[1] < hi]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return n.split("(")[0][:46]
small, big = collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gx, gy, gz, w in win:
    d = small if (gx // w) * gy * gz <= 260 else big
    d[short(n)][0] += 1; d[short(n)][1] += e - s
for title, d in (("SMALL-GRID (<= 260 workgroups)", small), ("LARGE-GRID", big)):
    print(f"{title}: {sum(v[1] for v in d.values()) / steps / 1e6:.2f} ms/step")
    for k, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"  {k:46s} n/step {c / steps:6.1f}  ms/step {t / steps / 1e6:6.3f}  avg {t / c / 1e3:6.1f} us")
