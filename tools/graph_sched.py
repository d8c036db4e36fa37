"""Offline model of the hipGraph executor's node -> stream assignment, checked against the StreamId fields of a DOT dump
(tools/graph_dot.py): python tools/graph_sched.py gpurun_out/step.dot"""
import collections
import re
import sys

sys.setrecursionlimit(100000)


def parse(path):
    text = open(path, errors="replace").read()
    nodes = {}
    order = []
    for m in re.finditer(r'"(graph_\d+_node_(\d+))"\[([^\]]*?)label="(\d+)\n([^\n]*)\n(.*?)"\];', text, re.S):
        key, idx, _, _, name, rest = m.groups()
        sid = int(re.search(r"StreamId:(\d+)", rest).group(1)) if "StreamId" in rest else -1
        nodes[key] = dict(idx=int(idx), name=name, sid=sid)
        order.append(key)
    edges = []
    for m in re.finditer(r'"(graph_\d+_node_\d+)"\s*->\s*"(graph_\d+_node_\d+)"', text):
        edges.append((m.group(1), m.group(2)))
    return nodes, order, edges


def simulate(nodes, order, edges, nq=4):
    out = collections.defaultdict(list); indeg = collections.Counter()
    for a, b in edges:
        out[a].append(b); indeg[b] += 1
    sid = {}

    def visit(n, s):
        if n in sid:
            return
        sid[n] = s
        for c in out[n]:
            visit(c, s)
            s = (s + 1) % nq
    s = 0
    for n in order:
        if indeg[n] == 0:
            visit(n, s)
            s = (s + 1) % nq
    return sid


def chains(path):
    """Queue ids (as the runtime assigned them) of the step's chains: the backbone's data-gradient chain, its weight gradients, the
    language branch's backward; and how many nodes the model of simulate() gets right."""
    nodes, order, edges = parse(path)
    sim = simulate(nodes, order, edges)
    idx = {n: nodes[n]["idx"] for n in order}
    name = lambda n: nodes[n]["name"]
    lang0 = min(idx[n] for n in order if "phrase_bwd" in name(n))
    lang1 = max(idx[n] for n in order if "embedding_bwd" in name(n))
    loss = min(idx[n] for n in order if "dense_loss_bwd" in name(n))
    in_lang = lambda n: lang0 <= idx[n] <= lang1
    # the backbone's backward: everything behind the head's backward that is not the language chain; its convolutions' data gradients
    # (conv3 / conv1 / igemm / dgrad2 / nconv kernels) against its weight-gradient kernels
    stem = max(idx[n] for n in order if "stem_wgrad" in name(n) or "wgrad9" in name(n))
    body = [n for n in order if loss + 300 < idx[n] <= stem and not in_lang(n)]
    is_w = lambda n: "wgrad" in name(n)
    is_d = lambda n: any(k in name(n) for k in ("conv3_kernel", "conv1_kernel", "dgrad2_kernel"))
    cnt = lambda ns: dict(collections.Counter(nodes[n]["sid"] for n in ns))
    return dict(agree=sum(1 for n in order if sim.get(n) == nodes[n]["sid"]), nodes=len(order),
                language=cnt([n for n in order if in_lang(n)]), data_gradients=cnt([n for n in body if is_d(n)]),
                weight_gradients=cnt([n for n in body if is_w(n)]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--chains":
        import json
        print(json.dumps(chains(sys.argv[1])))
        sys.exit(0)
    nodes, order, edges = parse(sys.argv[1])
    print("nodes", len(nodes), "edges", len(edges), "roots", sum(1 for n in order if not any(b == n for a, b in edges)))
    sim = simulate(nodes, order, edges)
    ok = sum(1 for n in order if sim.get(n) == nodes[n]["sid"])
    print("model agrees on %d of %d nodes" % (ok, len(order)))
    bad = [n for n in order if sim.get(n) != nodes[n]["sid"]][:10]
    for n in bad:
        print("  ", n, nodes[n]["name"][:60], "actual", nodes[n]["sid"], "model", sim.get(n))


def show(path, lo, hi):
    nodes, order, edges = parse(path)
    out = collections.defaultdict(list); inn = collections.defaultdict(list)
    for a, b in edges:
        out[a].append(b); inn[b].append(a)
    short = lambda k: int(k.split("_")[-1])
    for n in order:
        i = nodes[n]["idx"]
        if lo <= i < hi:
            nm = re.sub(r"^_ZN?\d*(_GLOBAL__N_1)?\d*", "", nodes[n]["name"])[:44]
            print("%5d s%d %-44s <- %-22s -> %s" % (i, nodes[n]["sid"], nm, [short(x) for x in inn[n]], [short(x) for x in out[n]]))
