"""How many fork / join points does a captured step graph have?  RF_GRAPH_DOT=<dir> makes the engine dump every captured
graph (hipGraphDebugDotPrint); this counts, per .dot file: nodes, edges, nodes with more than one successor (forks) and with
more than one predecessor (joins), and the width of the widest level.  python tools/graph_edges.py <dir>
(On this image -- PyTorch's bundled HIP 7.0.2 -- `CUDAGraph.debug_dump` warns and writes nothing: the directory stays empty.
The stand-alone probes under tools/probes/ print their own graphs with hipGraphDebugDotPrint; this script reads those.)"""
import glob, os, re, sys
from collections import defaultdict

for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.dot"))):
    succ, pred, nodes = defaultdict(set), defaultdict(set), set()
    label = {}
    for line in open(f, errors="replace"):
        m = re.search(r'"?([\w\d_]+)"?\s*->\s*"?([\w\d_]+)"?', line)
        if m:
            a, b = m.group(1), m.group(2)
            succ[a].add(b); pred[b].add(a); nodes.add(a); nodes.add(b)
            continue
        m = re.match(r'\s*"?([\w\d_]+)"?\s*\[.*label="([^"]*)"', line)
        if m:
            nodes.add(m.group(1)); label[m.group(1)] = m.group(2)
    forks = [n for n in nodes if len(succ[n]) > 1]
    joins = [n for n in nodes if len(pred[n]) > 1]
    edges = sum(len(v) for v in succ.values())
    # longest-path levels
    indeg = {n: len(pred[n]) for n in nodes}
    level = {n: 0 for n in nodes if indeg[n] == 0}
    order = [n for n in nodes if indeg[n] == 0]
    i = 0
    while i < len(order):
        n = order[i]; i += 1
        for m_ in succ[n]:
            level[m_] = max(level.get(m_, 0), level[n] + 1)
            indeg[m_] -= 1
            if indeg[m_] == 0:
                order.append(m_)
    width = defaultdict(int)
    for n, l in level.items():
        width[l] += 1
    print(f"{os.path.basename(f)}: {len(nodes)} nodes, {edges} edges, {len(forks)} forks, {len(joins)} joins, "
          f"critical path {max(level.values()) + 1 if level else 0} nodes, widest level {max(width.values()) if width else 0}")
    kinds = defaultdict(int)
    for n in forks:
        kinds[re.sub(r"[<(].*", "", label.get(n, n))[:48]] += 1
    print("   forks at:", dict(sorted(kinds.items(), key=lambda kv: -kv[1])[:12]))
