"""A short hash of the product's kernel and host sources.

Measurements that cannot be taken inside ``bench.py``'s own run (launch durations inside a replayed hipGraph, PMC byte counters:
``profiles/in_step_latest.json``, ``profiles/pmc_latest.json``) are stamped with the hash of the sources they were taken on;
``bench.py`` quotes them only while that hash is the one of the sources it runs, so a stale profile cannot pass as current.
"""
from __future__ import annotations

import hashlib
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_hash() -> str:
    """sha256 over csrc/* and the package's Python files (names and contents, sorted), first 12 hex digits."""
    h = hashlib.sha256()
    files = []
    for d in (os.path.join(_PKG, "csrc"), _PKG, os.path.join(_PKG, "utils")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h", ".cpp", ".py")):
                files.append(os.path.join(d, f))
    for path in files:
        h.update(os.path.relpath(path, _PKG).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    # the compile flags the library was built with (dcnet_amd/build.py writes the stamp): an experiment build
    # (DCN_EXTRA_FLAGS=-D..._ABL) must not carry the hash of the default build of the same sources
    try:
        with open(os.path.join(_PKG, "build", "FLAGS.stamp"), "rb") as fh:
            h.update(b"flags:" + fh.read().strip())
    except OSError:
        h.update(b"flags:unknown")
    return h.hexdigest()[:12]
