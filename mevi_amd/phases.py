"""Wall-clock phases of a CLI process, for the MS MARCO-sized rehearsal of the scripts (tools/e2e_cli.py).

The drop-in IS the command lines (MEVI/marco_eval_nci_rq.sh, MEVI/marco_ensemble.sh): what a user waits for is interpreter
start-up + imports + file reads + uploads + kernels + file writes of each fresh process.  With MEVI_PHASE_LOG=<file> every
`mark(name)` appends one line `script<TAB>phase<TAB>seconds of the phase<TAB>seconds since the process was created`; the
first mark therefore carries interpreter start-up and imports.  Unset (the default) a mark is a no-op."""
import os
import sys
import time

_LOG = os.environ.get("MEVI_PHASE_LOG")
_last = None


def _process_start():
    try:
        import psutil

        return psutil.Process().create_time()
    except Exception:       # no psutil: count from the first mark
        return time.time()


def mark(name, sync=False):
    """End the current phase under `name`.  sync=True waits for the GPU first (phases that end with queued kernels)."""
    global _last
    if not _LOG:
        return
    if sync:
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
    now = time.time()
    if _last is None:
        _last = _START
    with open(_LOG, "a") as f:
        f.write("%s\t%s\t%.3f\t%.3f\n" % (os.path.basename(sys.argv[0]) or "python", name, now - _last, now - _START))
    _last = now


_START = _process_start() if _LOG else 0.0


def finish(code=0):
    """Leave a single-process CLI run NOW: flush the standard streams and `os._exit` -- skipping the interpreter's teardown,
    which for these scripts means returning 40+ GB of device memory and the pinned staging ring piece by piece and shutting the
    HIP runtime down (0.4-1.2 s per process in tools/e2e_cli.py) for a process whose memory the driver reclaims at exit anyway.
    Every output file is closed (or flushed, for memory maps) by the code that wrote it before this is called.
    MEVI_FAST_EXIT=0 keeps the ordinary exit."""
    mark("done")
    if os.environ.get("MEVI_FAST_EXIT", "1") == "0":
        return
    try:
        sys.stdout.flush()
        sys.stderr.flush()
    finally:
        os._exit(code)
