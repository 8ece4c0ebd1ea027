"""CPU-side checks: the C-ABI library loads and exports every symbol include/mevi_hip.h
declares; host-only entry points behave; the product path refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mevi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mevi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from mevi_amd import hip
    from mevi_amd.build import build

    build()
    L = hip.lib()
    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/mevi_hip.h"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in mevi_hip.h but not exported"
    assert sorted(hip.exported_symbols()) == declared
    assert L.mevi_abi_version() == 1


def test_workspace_queries_are_pure_host_arithmetic():
    from mevi_amd import hip

    L = hip.lib()
    a = L.mevi_ip_topk_workspace_bytes(6980, 768, 1000)
    assert a > 2 * 6980 * 4096 * 8
    assert L.mevi_ip_topk_workspace_bytes(6980, 768, 5000) == 0   # k > 4096 unsupported
    assert L.mevi_ip_topk_workspace_bytes(0, 768, 10) == 0


def test_product_path_fails_loudly_without_gpu():
    import torch

    from mevi_amd import dense, hip

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.MeviHipError):
        dense.search(np.zeros((2, 8), np.float32), np.zeros((4, 8), np.float32), 8, 2, "Flat")


def test_product_never_imports_oracle():
    """The product path must not import, link, dlopen or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "mevi_amd")
    cli = [os.path.join(ROOT, f) for f in os.listdir(ROOT)
           if f.endswith(".py") and f not in ("bench.py", "__graft_entry__.py")]
    files = cli[:]
    for dirpath, _, fs in os.walk(pkg):
        files += [os.path.join(dirpath, f) for f in fs if f.endswith((".py", ".hip", ".h", ".cpp"))]
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|#include\s+[\"<][^\">]*oracle|libmevi_oracle|dlopen", re.M)
    for f in files:
        assert not pat.search(open(f).read()), f
