"""Build libdcnet_hip.so (the C-ABI HIP kernel library) in-tree with hipcc for gfx950.

    python -m dcnet_amd.build            # incremental
    python -m dcnet_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdcnet_hip.so")
SOURCES = ["igemm.hip", "conv3.hip", "conv3x.hip", "conv1.hip", "nconv.hip", "stem.hip", "conv.hip", "wgrad.hip", "wgrad3.hip", "wgrad9.hip", "bn.hip", "layout.hip", "coattn.hip", "gemm3.hip", "score.hip", "gemm.hip", "locmod.hip", "optim.hip", "wprep.hip", "phrase.hip", "head.hip", "sample.hip", "loss.hip", "lstm.hip", "fusion.hip", "post.hip", "b16.hip", "conv2b.hip",
           "sampling.cpp", "capi.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Werror=array-bounds"]
FLAGS += os.environ.get("DCN_EXTRA_FLAGS", "").split()      # experiment builds, e.g. -DC3_ABL=1 (timing ablations of conv3.hip)
if os.environ.get("DCN_WGRAD_KP"):          # experiment knob: pixels per K-step of the weight-gradient kernel
    FLAGS.append("-DWGRAD_KP=" + os.environ["DCN_WGRAD_KP"])


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def flags_key() -> str:
    """Hash of the compile flags: part of the object-cache key and of ``utils.srchash.kernel_sources_hash`` — an experiment
    build (``DCN_EXTRA_FLAGS=-DC3_ABL=1`` …) can neither be mistaken for an up-to-date default build nor stamp a bench line
    with the default build's source hash."""
    import hashlib
    return hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:12]


def built_flags_key() -> str:
    """The flags key of the objects libdcnet_hip.so was linked from ('' if never built by this script)."""
    try:
        with open(os.path.join(HERE, "build", "FLAGS.stamp")) as fh:
            return fh.read().strip()
    except OSError:
        return ""


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "dcnet_hip.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "FLAGS.stamp")
    if built_flags_key() != flags_key():        # other flags than the objects were compiled with: every object is stale
        force = True
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        if force or _stale(obj, [src] + headers):
            extra = ["-x", "hip"] if s.endswith(".hip") else []
            jobs.append((s, [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return name

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            for name in ex.map(run, jobs):
                if verbose:
                    print("compiled", name)
    objs = [os.path.join(objdir, s.rsplit(".", 1)[0] + ".o") for s in srcs]
    if force or jobs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("linked", OUT)
    with open(stamp, "w") as fh:
        fh.write(flags_key() + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
