"""The captured training step as a DAG (GPU box): python tools/graph_dot.py [--out gpurun_out/step.dot] [--from bilstm_bwd] [--to dA_kernel]
Dumps the hipGraph of one captured step (DEBUG_HIP_GRAPH_DOT_PRINT) and prints, for the first kernel node whose name holds --from, how many
nodes depend on it and the shortest dependency path to each kernel whose name holds --to."""
import argparse
import collections
import os
import random
import re
import sys

os.environ.setdefault("DEBUG_HIP_GRAPH_DOT_PRINT", "1")      # the runtime writes graph_<pid>_dot_print_<n> into the working directory at instantiation

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def capture(path, clips, precision, switches="", size=416):
    from dcnet_amd import ops
    for kv in switches.split(","):
        if kv:
            setattr(ops, kv.split("=")[0], int(kv.split("=")[1]))
    from dcnet_amd.graph import GraphedTrainStep
    from dcnet_amd.model import grounding_model
    from dcnet_amd.parallel import freeze_gradless
    from dcnet_amd.train import make_optimizer
    from dcnet_amd.utils.synth import synth_boxes, synth_inputs
    dev = torch.device("cuda:0")
    ops.set_precision(precision)
    torch.manual_seed(1234)
    model = grounding_model(corpus=list(range(1000)), light=False, emb_size=512, coordmap=True, bert_model="bert-base-uncased",
                            dataset="vid", img_size=size, config_path=os.path.join(ROOT, "model", "yolov3.cfg"), weights_path=None).to(dev)
    model.train(); freeze_gradless(model)
    opt = make_optimizer(model, 1e-4)
    n = clips * 8
    image, word_id, word_mask = (t.to(dev) for t in synth_inputs(n, size, seed=100))
    bbox = synth_boxes(n, size, seed=100).to(dev)
    random.seed(13)
    cwd = os.getcwd(); os.chdir("/tmp")
    GraphedTrainStep(model, opt, image, word_id, word_mask, bbox, size, warmup=2)
    torch.cuda.synchronize()
    os.chdir(cwd)
    import glob
    import shutil
    dots = sorted(glob.glob("/tmp/graph_%d_dot_print_*" % os.getpid()), key=os.path.getsize)
    shutil.copy(dots[-1], path)


def analyse(path, src, dst):
    text = open(path, errors="replace").read()
    label = {}
    for m in re.finditer(r'"?([A-Za-z0-9_]+)"?\s*\[([^\]]*)\]', text):
        lab = re.search(r'label\s*=\s*"((?:[^"\\]|\\.)*)"', m.group(2), re.S)
        if lab:
            label[m.group(1)] = lab.group(1).replace("\\n", " ")[:140]
    out = collections.defaultdict(list)
    n_edges = 0
    for m in re.finditer(r'"?([A-Za-z0-9_]+)"?\s*->\s*"?([A-Za-z0-9_]+)"?', text):
        out[m.group(1)].append(m.group(2)); n_edges += 1
    print("nodes %d edges %d" % (len(label), n_edges))
    starts = [k for k, v in label.items() if src in v]
    if not starts:
        print("no node matches", src); return
    s = starts[0]
    print("from:", s, label[s])
    prev = {s: None}; q = collections.deque([s])
    while q:
        a = q.popleft()
        for b in out[a]:
            if b not in prev:
                prev[b] = a; q.append(b)
    print("descendants: %d of %d nodes" % (len(prev) - 1, len(label)))
    shown = 0
    for k, v in label.items():
        if dst in v and k in prev and k != s:
            path_ = []; a = k
            while a is not None:
                path_.append(a); a = prev[a]
            print("---- path to", k, "(%d hops)" % (len(path_) - 1))
            for a in reversed(path_):
                print("     ", a, label.get(a, "?"))
            shown += 1
            if shown >= 2:
                break
    if not shown:
        print("no node matching %r depends on it" % dst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "step.dot"))
    ap.add_argument("--from", dest="src", default="bilstm_bwd"); ap.add_argument("--to", dest="dst", default="dA_kernel")
    ap.add_argument("--clips", type=int, default=8); ap.add_argument("--precision", default="fp32")
    ap.add_argument("--analyse-only", action="store_true")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--ops", default="", help="switches of dcnet_amd.ops for the capture, NAME=int[,NAME=int]")
    a = ap.parse_args()
    if not a.analyse_only:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        capture(a.out, a.clips, a.precision, a.ops, a.size)
    analyse(a.out, a.src, a.dst)


if __name__ == "__main__":
    main()
