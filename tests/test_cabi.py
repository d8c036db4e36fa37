"""CPU: the C-ABI library loads and exports every symbol that include/dcnet_hip.h declares, the
ctypes table covers them all, and the product refuses to run without the GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from util import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "dcnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from dcnet_amd.lib import LIB_PATH, SIGNATURES, lib
    names = _declared()
    assert len(names) >= 35
    assert os.path.exists(LIB_PATH), "build the library first: python -m dcnet_amd.build"
    dll = ctypes.CDLL(LIB_PATH)
    for n in names:
        assert hasattr(dll, n), f"{n} declared in dcnet_hip.h but not exported"
        assert n in SIGNATURES, f"{n} has no ctypes signature in dcnet_amd/lib.py"
    assert set(SIGNATURES) <= set(names), sorted(set(SIGNATURES) - set(names))
    L = lib()
    from dcnet_amd.lib import ABI_VERSION
    assert L.version() == ABI_VERSION
    assert L.conv2d_stats_rows(2, 13, 13, 128, 3, 1) in (3, 6)     # 338 rows in 128- or 64-row tiles: pure host arithmetic
    assert L.channel_stats_rows(1000) == 8
    assert L.conv2d_geom_size(2, 13, 13, 3, 1) == 2 * 13 * 13 + 128          # one entry per output pixel + prefetch slack
    assert L.conv2d_geom_size(1, 416, 416, 3, 2) == 208 * 208 + 128
    assert L.coattn_e_size(2, 169) == 2 * 169 * 192


def test_bad_arguments_are_rejected_with_a_message():
    from dcnet_amd.lib import DcnError, lib
    L = lib()
    with pytest.raises(DcnError) as e:
        L.conv2d_fwd(0, 0, 0, 1, 8, 8, 33, 16, 3, 1, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)     # cin not a multiple of 32
    assert "cin" in str(e.value)
    with pytest.raises(DcnError):
        L.conv2d_fwd(0, 0, 0, 1, 8, 8, 32, 16, 5, 1, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)     # ksize 5


def test_product_has_no_cpu_path():
    from util import build_product, synth_sd
    m = build_product(256, synth_sd(256), torch.device("cpu")).eval()
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(2, 3, 256, 256), torch.ones(2, 20, dtype=torch.long), None)


def test_product_never_imports_the_oracle():
    bad = []
    for d in ("dcnet_amd", "model"):
        for root, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h")):
                    src = open(os.path.join(root, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(root, f))
    assert not bad, bad
