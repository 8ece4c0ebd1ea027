"""CPU side of bench.py's `cpu_baseline` (VERDICT r2 #3b): the dense arm the way the reference runs it on host cores.

Order of preference:
  1. `faiss` itself (faiss-cpu IndexFlatIP, MEVI/faiss_search.py:13-21) when importable on the box  -> kind "reference";
  2. the port `oracle.dense.ip_topk_blas` (blocked sgemm + per-query heaps = how faiss evaluates Flat-IP), with the
     sgemm backend, thread count and block size SWEPT on a small calibration problem and the best combination used for the
     bounded sample -> kind "port".  Backends: torch.mm (MKL in this image) and numpy (OpenBLAS); threads {32, 64, 128}
     (those the box has), pinned to distinct physical cores first; blocks {1024 = faiss's distance_compute_blas_database_bs,
     16384}.
Only bench.py (cpu_baseline leg) imports this; it is the only place besides tests/ that touches oracle/."""
import os
import time

import numpy as np


def physical_core_cpus():
    """One logical CPU per physical core first, then the hyper-thread siblings (so that the first n entries are n
    distinct cores whenever n <= cores)."""
    first, rest, seen = [], [], set()
    allowed = sorted(os.sched_getaffinity(0))
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = f.read().strip()
        except OSError:
            sib = str(c)
        if sib in seen:
            rest.append(c)
        else:
            seen.add(sib)
            first.append(c)
    return first + rest, len(first)


def pin_all_threads(cpus):
    """Affinity of every thread of this process (BLAS / OpenMP pools are created with the process's mask)."""
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cpus)
        except (OSError, ValueError):
            pass


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _sgemm(backend):
    if backend == "torch":
        import torch

        def mm(q, d):
            return torch.mm(torch.from_numpy(q), torch.from_numpy(d).t()).numpy()
        return mm
    return lambda q, d: np.ascontiguousarray(q @ d.T)


def _blas_search(odense, q, d, k, block, backend, timing=None):
    """oracle.dense.ip_topk_blas with a selectable sgemm backend (same heaps, same tie rule)."""
    from ctypes import c_void_p

    L = odense.lib()
    nq = q.shape[0]
    heap_s = np.empty((nq, k), np.float32)
    heap_i = np.empty((nq, k), np.int64)
    heap_n = np.zeros(nq, np.int64)
    mm = _sgemm(backend)
    p = lambda a: a.ctypes.data_as(c_void_p)      # noqa: E731
    tg = th = 0.0
    for b0 in range(0, d.shape[0], block):
        t0 = time.perf_counter()
        sc = mm(q, d[b0:b0 + block])
        t1 = time.perf_counter()
        L.oracle_heap_update_f32(p(sc), nq, sc.shape[1], b0, k, p(heap_s), p(heap_i), p(heap_n))
        tg, th = tg + (t1 - t0), th + (time.perf_counter() - t1)
    t1 = time.perf_counter()
    L.oracle_heap_finalize_f32(nq, k, p(heap_s), p(heap_i), p(heap_n))
    th += time.perf_counter() - t1
    if timing is not None:
        timing["sgemm_s"], timing["heap_s"] = tg, th
    return heap_s, heap_i


def dense_baseline(n_docs, nq_full, dim, topk, target_s=15.0, seed=7):
    """queries/s of the CPU dense arm at the full corpus size, from a bounded sample scaled linearly in rows."""
    import threadpoolctl

    from oracle import dense as odense

    rng = np.random.default_rng(seed)
    nd_s = min(n_docs, 500_000)
    d = (0.05 * rng.standard_normal((nd_s, dim), dtype=np.float32) + 0.02).astype(np.float32)
    q_all = (0.05 * rng.standard_normal((min(nq_full, 4096), dim), dtype=np.float32) + 0.02).astype(np.float32)
    host_cpus = os.cpu_count()
    order, n_phys = physical_core_cpus()
    out = {"unit": "queries/s", "host_cpus": host_cpus, "physical_cores": n_phys, "cgroup_cpu_max": _read("/sys/fs/cgroup/cpu.max"),
           "loadavg": _read("/proc/loadavg")}
    quota = None
    try:        # "max 100000" = unlimited; "1600000 100000" = 16 CPUs' worth of time whatever the number of threads
        q_, per_ = (out["cgroup_cpu_max"] or "max 1").split()[:2]
        quota = None if q_ == "max" else float(q_) / float(per_)
    except ValueError:
        pass
    out["cgroup_quota_cpus"] = quota

    # ---- 1. faiss itself ------------------------------------------------------------------------------------------------
    try:
        import faiss      # noqa: F401  (not in this image; BASELINE.md section 3 asks to try)

        index = faiss.index_factory(dim, "Flat", faiss.METRIC_INNER_PRODUCT)
        index.add(d)
        index.search(q_all[:64], topk)
        nq_s = min(len(q_all), 1024)
        t = time.perf_counter()
        index.search(q_all[:nq_s], topk)
        dt = time.perf_counter() - t
        out.update({"value": nq_s / dt * nd_s / n_docs, "cores": faiss.omp_get_max_threads(), "kind": "reference",
                    "backend": "faiss-cpu IndexFlatIP",
                    "sample": f"faiss {faiss.__version__}: {nq_s} queries x {nd_s} docs x {dim}, top-{topk} in {dt:.2f}s; "
                              f"scaled x{nd_s}/{n_docs} rows to the full corpus"})
        return out
    except Exception as e:      # ImportError here; anything else is reported and the port runs
        out["faiss"] = f"not usable on this box ({type(e).__name__}: {e}); port = blocked sgemm + heaps"

    # ---- 2. the port, swept ----------------------------------------------------------------------------------------------
    all_cpus = set(os.sched_getaffinity(0))
    cal_q, cal_d = q_all[:256], d[:32768]
    sweep = []
    threads_opts = sorted({t for t in (8, 16, 32, 64, 128) if t <= len(order)} | ({min(len(order), n_phys)} if n_phys <= 128 else set()))
    best = None
    for backend in ("torch", "numpy"):
        for nt in threads_opts:
            pin_all_threads(set(order[:nt]))
            try:
                import torch

                torch.set_num_threads(nt)
            except Exception:
                pass
            with threadpoolctl.threadpool_limits(limits=nt):
                for block in (1024, 16384):
                    _blas_search(odense, cal_q[:64], cal_d[:8192], topk, block, backend)       # warm the pools
                    tm = {}
                    t = time.perf_counter()
                    _blas_search(odense, cal_q, cal_d, topk, block, backend, tm)
                    dt = time.perf_counter() - t
                    row = {"backend": backend, "threads": nt, "block": block, "s": round(dt, 3),
                           "sgemm_tflops": round(2.0 * len(cal_q) * len(cal_d) * dim / max(tm["sgemm_s"], 1e-9) / 1e12, 3)}
                    sweep.append(row)
                    if best is None or dt < best[0]:
                        best = (dt, backend, nt, block)
    _, backend, nt, block = best
    pin_all_threads(set(order[:nt]))
    try:
        import torch

        torch.set_num_threads(nt)
    except Exception:
        pass
    with threadpoolctl.threadpool_limits(limits=nt):
        t = time.perf_counter()
        _blas_search(odense, q_all[:256], d, topk, block, backend)
        cal = time.perf_counter() - t
        nq_s = int(min(len(q_all), max(256, 256 * target_s / max(cal, 1e-3))))
        tm = {}
        t = time.perf_counter()
        _blas_search(odense, q_all[:nq_s], d, topk, block, backend, tm)
        dt = time.perf_counter() - t
    pin_all_threads(all_cpus)
    try:
        import torch

        torch.set_num_threads(min(host_cpus, len(all_cpus)))
    except Exception:
        pass
    out.update({
        "value": nq_s / dt * nd_s / n_docs, "cores": nt if quota is None else min(nt, int(round(quota))), "threads": nt, "kind": "port",
        "cores_note": None if quota is None else f"{nt} threads, but the box's cgroup grants {quota:g} CPUs of time (cpu.max): that is what "
                                                 f"bounds the sgemm rate, not the {n_phys} physical cores",
        "backend": f"{backend} sgemm ({'MKL' if backend == 'torch' else 'OpenBLAS'}), block {block}, {nt} threads pinned to "
                   f"{min(nt, n_phys)} physical cores" + (" + siblings" if nt > n_phys else ""),
        "sample": f"{nq_s} queries x {nd_s} docs x {dim} f32, top-{topk}: blocked sgemm + per-query heaps (faiss Flat-IP "
                  f"algorithm, oracle.dense heaps) took {dt:.2f}s; scaled x{nd_s}/{n_docs} rows to the full corpus",
        "sgemm_s": tm["sgemm_s"], "heap_s": tm["heap_s"],
        "sgemm_tflops": 2.0 * nq_s * nd_s * dim / max(tm["sgemm_s"], 1e-9) / 1e12,
        "sweep": sweep,
    })
    return out


def dense_baseline_subprocess(n_docs, nq_full, dim, topk, target_s=15.0, timeout_s=240):
    """dense_baseline in a CHILD process: the sweep pins threads and resizes the BLAS / OpenMP pools, which must not leak
    into the process that goes on to time the GPU path and the oracle legs (round 3, first run: the oracle's NCI generate
    took 1356 s instead of ~20 s with the pools left oversubscribed)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench_cpu; "
            "print('CPU_BASELINE_JSON ' + json.dumps(bench_cpu.dense_baseline(%d, %d, %d, %d, target_s=%r)))"
            % (root, os.path.join(root, "tools"), n_docs, nq_full, dim, topk, target_s))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")      # the child never touches the GPU
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
    for line in r.stdout.splitlines():
        if line.startswith("CPU_BASELINE_JSON "):
            return json.loads(line[len("CPU_BASELINE_JSON "):])
    raise RuntimeError("cpu baseline child failed: " + (r.stderr or r.stdout)[-400:])
