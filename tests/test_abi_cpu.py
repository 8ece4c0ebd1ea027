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
    # the image width the host computes without a foreign call is the library's
    from mevi_amd import ops

    for k in list(range(1, 300)) + [767, 768, 769, 2048, 3072, 3073, 100000]:
        assert ops.split_kp(k) == L.mevi_split_kp(k), k


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


def test_bench_starts_its_ranks_itself_and_reports_their_failure():
    """`python bench.py --gpus 2` outside a launcher: the parent starts two ranks through
    torch.distributed.run as a child process -- it never imports torch itself, so it cannot have touched the GPU -- and
    hands their exit status on.  Without a GPU both ranks fail loudly (no CPU fallback), so the status is non-zero."""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: the GPU suite runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("no MI355X visible") >= 2, r.stderr[-1500:]          # both ranks got as far as the GPU check
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def late_imports")]
    assert "import torch" not in head and "mevi_amd" not in head.split('"""', 2)[2]   # nothing GPU-capable before the spawn


def test_bench_line_is_compact_and_keeps_the_certificate_in_config():
    """The ONE stdout line must fit the driver's 8 KB tail and carry the MRR@10-match certificate and the chain's rate in
    `config`: checked on a recorded full record."""
    import contextlib
    import io
    import json
    import tempfile

    import bench

    with open(os.path.join(ROOT, "profiles", "r03_bench_n1.json")) as f:
        rec = json.load(f)
    old = os.environ.get("MEVI_BENCH_DETAIL")
    os.environ["MEVI_BENCH_DETAIL"] = os.path.join(tempfile.mkdtemp(), "d.json")
    try:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.print_line(rec)
        with open(os.environ["MEVI_BENCH_DETAIL"]) as f:
            full = json.load(f)
    finally:
        if old is None:
            del os.environ["MEVI_BENCH_DETAIL"]
        else:
            os.environ["MEVI_BENCH_DETAIL"] = old
    out = buf.getvalue()
    assert out.count("\n") == 1 and len(out) < 8192
    line = json.loads(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("C2") and line["roofline"]["kernel"] == "ip_filter_h1_kernel"
    m = line["config"]["mrr10_match"]
    assert m["dense_lists_identical"] is True and m["beams_identical"] is True and m["chain_mrr10_abs_diff"] == 0.0
    assert line["config"]["chain_c4"]["queries_per_s"] > 0
    assert "per_batch" in full["dense_small_batch"] and "per_batch" not in line.get("dense_small_batch", {})
