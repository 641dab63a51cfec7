"""Reads a graph written by hipGraphDebugDotPrint (tools/probes/streams_race_probe.py --dot): node names, edges, and for a range
of node ids the nodes each one is UNORDERED against (neither an ancestor nor a descendant: what the runtime may run at the same
time).  Used for DESIGN 7 #5: in the recorded batch of a failing stream deal no node of the generator's backward pass is unordered
against anything.
  python tools/probes/graph_dot_check.py g.2.dot [first_id last_id]"""
import collections
import re
import sys


def load(path):
    s = open(path).read()
    names = {}
    for m in re.finditer(r'"graph_\d+_node_(\d+)"\[', s):
        n = int(m.group(1))
        blk = s[m.end():m.end() + 600]
        k = re.search(r"ID \| \d+ \| ([^}]*)\}", blk)
        if k:
            k = k.group(1)
            mm = re.search(r"GLOBAL__N_\d+(\w+?)(?:E[vPK]|I[LN])", k) or re.search(r"native\d+(\w+?)I", k)
            g = re.search(r"<<<\(([\d,]+)\)", k.replace("\\", ""))
            names[n] = (mm.group(1) if mm else k[:30]) + ("(%s)" % g.group(1) if g else "")
        else:
            names[n] = blk.split("|")[0].strip().strip("{").strip()[:10]  # MEMCPY / MEMSET nodes
    edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_\d+_node_(\d+)"\s*->\s*"graph_\d+_node_(\d+)"', s)]
    return names, edges


def main():
    names, edges = load(sys.argv[1])
    ids = sorted(names)
    idx = {n: i for i, n in enumerate(ids)}
    pred, succ = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in edges:
        pred[b].append(a)
        succ[a].append(b)
    assert all(a < b for a, b in edges), "node ids are not a topological order"
    anc = {}
    for n in ids:  # ancestors as bit sets
        bits = 0
        for q in pred[n]:
            bits |= anc[q] | (1 << idx[q])
        anc[n] = bits
    kinds = collections.Counter(v.split("(")[0] for v in names.values())
    print("%d nodes, %d edges; joins (in-degree > 1): %d, forks (out-degree > 1): %d" % (
        len(ids), len(edges), sum(len(pred[n]) > 1 for n in ids), sum(len(succ[n]) > 1 for n in ids)))
    print("most frequent:", ", ".join("%s x %d" % kv for kv in kinds.most_common(6)))
    lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (ids[0], ids[-1])
    for n in ids:
        if not lo <= n <= hi:
            continue
        un = [m for m in ids if m < n and not (anc[n] >> idx[m]) & 1] + [m for m in ids if m > n and not (anc[m] >> idx[n]) & 1]
        print(n, names[n], "<-", sorted(pred[n]), "| unordered against", len(un), un[:6] + (["..."] if len(un) > 6 else []))


if __name__ == "__main__":
    main()
