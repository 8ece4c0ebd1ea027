"""Where the host -> HBM upload of a mapped corpus file loses its rate: fill rates into the different kinds of
staging memory without any device copy, DMA rates out of them, and a mapping registered in place (no CPU copy at all)."""
import ctypes
import mmap
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

dev = torch.device("cuda:0")
torch.cuda.init()
rt = torch.cuda.cudart()
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
nbytes = int(GB * (1 << 30)) // (1 << 21) * (1 << 21)
d = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
path = os.path.join(d, "mevi_probe_upload2.bin")
np.random.default_rng(0).integers(0, 255, nbytes, dtype=np.uint8).tofile(path)
print("file", path, nbytes / 1e9, "GB; cpus", len(os.sched_getaffinity(0)),
      "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", flush=True)
fd = os.open(path, os.O_RDONLY)


def rate(t):
    return nbytes / t / 1e9


def pread_into(mv, off):
    got = 0
    while got < len(mv):
        n = os.preadv(fd, [mv[got:]], off + got)
        assert n > 0
        got += n


def fill_threads(buf_u8, th, piece=1 << 24):
    """buf_u8: numpy uint8 view of the destination (nbytes); pread the file into it from `th` threads."""
    mv = memoryview(buf_u8)
    offs = list(range(0, nbytes, piece))
    with ThreadPoolExecutor(th) as pool:
        t = time.perf_counter()
        list(pool.map(lambda o: pread_into(mv[o:o + piece], o), offs))
        return time.perf_counter() - t


try:
    # 1. fill only
    plain = np.empty(nbytes, np.uint8)
    plain[:] = 0
    for th in (4, 8, 16):
        print("fill plain numpy        %2d threads: %.1f GB/s" % (th, rate(fill_threads(plain, th))), flush=True)
    pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    pv = pinned.numpy()
    for th in (4, 8, 16):
        print("fill torch pinned       %2d threads: %.1f GB/s" % (th, rate(fill_threads(pv, th))), flush=True)
    src2 = np.empty(nbytes, np.uint8)
    src2[:] = 1
    for th in (4, 8, 16):
        piece = 1 << 24
        with ThreadPoolExecutor(th) as pool:
            t = time.perf_counter()
            list(pool.map(lambda o: np.copyto(pv[o:o + piece], src2[o:o + piece]), range(0, nbytes, piece)))
            dt = time.perf_counter() - t
        print("memcpy plain -> pinned  %2d threads: %.1f GB/s" % (th, rate(dt)), flush=True)
        with ThreadPoolExecutor(th) as pool:
            t = time.perf_counter()
            list(pool.map(lambda o: np.copyto(plain[o:o + piece], src2[o:o + piece]), range(0, nbytes, piece)))
            dt = time.perf_counter() - t
        print("memcpy plain -> plain   %2d threads: %.1f GB/s" % (th, rate(dt)), flush=True)
    out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        print("DMA torch pinned -> device: %.1f GB/s" % rate(time.perf_counter() - t), flush=True)
    # 2. registered plain memory
    r = rt.cudaHostRegister(plain.ctypes.data, nbytes, 0)
    print("cudaHostRegister(plain) ->", r, flush=True)
    if int(r) == 0:
        tp = torch.from_numpy(plain)
        for _ in range(2):
            torch.cuda.synchronize()
            t = time.perf_counter()
            rt.cudaMemcpyAsync if False else None
            libhip = ctypes.CDLL("libamdhip64.so")
            libhip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
            rc = libhip.hipMemcpyAsync(out.data_ptr(), plain.ctypes.data, nbytes, 1, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            print("DMA registered plain -> device: rc %d %.1f GB/s" % (rc, rate(time.perf_counter() - t)), flush=True)
        for th in (8, 16):
            print("fill registered plain   %2d threads: %.1f GB/s" % (th, rate(fill_threads(plain, th))), flush=True)
        rt.cudaHostUnregister(plain.ctypes.data)
    # 3. the mapping itself registered (no CPU copy): read-only shared, then private writable
    libhip = ctypes.CDLL("libamdhip64.so")
    libhip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    libhip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    libhip.hipHostUnregister.argtypes = [ctypes.c_void_p]
    for name, prot, flags, regflags in (("shared read-only", mmap.PROT_READ, mmap.MAP_SHARED, 0),
                                        ("shared read-only, hipHostRegisterReadOnly", mmap.PROT_READ, mmap.MAP_SHARED, 8),
                                        ("private writable", mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE, 0)):
        mm = mmap.mmap(fd, nbytes, flags=flags, prot=prot)
        arr = np.frombuffer(mm, np.uint8)
        addr = arr.ctypes.data
        win = 1 << 28
        t = time.perf_counter()
        ok, treg = True, 0.0
        for o in range(0, nbytes, win):
            n = min(win, nbytes - o)
            t1 = time.perf_counter()
            rc = libhip.hipHostRegister(addr + o, n, regflags)
            treg += time.perf_counter() - t1
            if rc != 0:
                print("mapping %s: hipHostRegister rc %d" % (name, rc), flush=True)
                ok = False
                break
            rc = libhip.hipMemcpyAsync(out.data_ptr() + o, addr + o, n, 1, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            libhip.hipHostUnregister(addr + o)
            if rc != 0:
                print("mapping %s: hipMemcpyAsync rc %d" % (name, rc), flush=True)
                ok = False
                break
        if ok:
            dt = time.perf_counter() - t
            same = bool((out[:1 << 20].cpu().numpy() == arr[:1 << 20]).all())
            print("mapping %s registered in 256 MB windows -> device: %.1f GB/s (register %.2f s of %.2f s) same=%s"
                  % (name, rate(dt), treg, dt, same), flush=True)
        del arr
        try:
            mm.close()
        except BufferError:
            pass
    # 4. pipelined: N pinned buffers, T threads pread + async copies (what upload_rows does), by chunk size
    for chunk_mb, nbuf, th in ((64, 4, 8), (64, 4, 16), (256, 3, 16), (16, 8, 16)):
        chunk = chunk_mb << 20
        bufs = [torch.empty(chunk, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
        views = [memoryview(b.numpy()) for b in bufs]
        evs = [torch.cuda.Event() for _ in range(nbuf)]
        stream = torch.cuda.current_stream()
        sub = max(1 << 20, chunk // th)
        with ThreadPoolExecutor(th) as pool:
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i, o in enumerate(range(0, nbytes, chunk)):
                n = min(chunk, nbytes - o)
                b = i % nbuf
                if i >= nbuf:
                    evs[b].synchronize()
                list(pool.map(lambda s: pread_into(views[b][s:min(s + sub, n)], o + s), range(0, n, sub)))
                out[o:o + n].copy_(bufs[b][:n], non_blocking=True)
                evs[b].record(stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        print("pipelined pread: chunk %d MB x %d buffers, %d threads: %.1f GB/s" % (chunk_mb, nbuf, th, rate(dt)), flush=True)
        del bufs, views
finally:
    os.close(fd)
    os.remove(path)
