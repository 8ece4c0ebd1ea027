"""Host -> HBM upload of a mapped corpus file: pread into the pinned buffers vs memcpy out of the mapping, by thread count."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import io as mio
dev = torch.device("cuda:0")
rows, dim = (2 << 30) // (4 * 768), 768
d = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
path = os.path.join(d, "mevi_probe_upload.bin")
np.random.default_rng(0).standard_normal((rows, dim), dtype=np.float32).tofile(path)
print("cpus:", len(os.sched_getaffinity(0)), "cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
try:
    for mode in ("pread", "copy"):
        os.environ["MEVI_UPLOAD"] = mode
        for th in (4, 8, 16, 32):
            m = mio.map_rows(path, dim)
            mio.upload_rows(m[:rows // 8], dev, threads=th)
            t = time.perf_counter()
            mio.upload_rows(m, dev, threads=th)
            torch.cuda.synchronize()
            print(mode, th, "threads: %.1f GB/s" % (rows * dim * 4 / (time.perf_counter() - t) / 1e9))
    host = np.fromfile(path, dtype=np.float32).reshape(rows, dim)
    for th in (8, 16):
        t = time.perf_counter(); mio.upload_rows(host, dev, threads=th); torch.cuda.synchronize()
        print("in-memory array", th, "threads: %.1f GB/s" % (rows * dim * 4 / (time.perf_counter() - t) / 1e9))
    pin = torch.from_numpy(host).pin_memory()
    t = time.perf_counter(); pin.to(dev, non_blocking=True); torch.cuda.synchronize()
    print("already pinned -> device: %.1f GB/s" % (rows * dim * 4 / (time.perf_counter() - t) / 1e9))
finally:
    os.remove(path)
