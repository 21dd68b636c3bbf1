"""Named stage timers with the reference's API (/root/reference/utils/timer.py:8-111), on HIP events.

    timer.cuda(); timer.reset()
    with timer.timer('Network Forward'): ...
    timer.get_all_elapsed_time()  ->  {'Network Forward': mean ms, ...}

``torch.cuda.Event`` on ROCm is a hipEvent recorded on torch's current stream -- the stream every kernel of
this package is launched on -- so the figures are device times, as in the reference's CUDATimer (:70-82).
"""
import time
from collections import OrderedDict

import torch

_is_cuda = True
_timer_history = OrderedDict()


def cuda():
    global _is_cuda
    _is_cuda = True


def cpu():
    global _is_cuda
    _is_cuda = False


def reset():
    _timer_history.clear()


def get_all_elapsed_time():
    """Mean milliseconds per timer name over all its uses (one device synchronisation)."""
    if _is_cuda:
        torch.cuda.synchronize()
    return {k: sum(t.elapsed_time() for t in v) / len(v) for k, v in _timer_history.items() if v}


def log_elapsed_time(logger=None):
    lines = ["%-28s %10s %10s" % ("Item", "Time (ms)", "FPS")]
    for k, v in get_all_elapsed_time().items():
        lines.append("%-28s %10.2f %10.2f" % (k, v, 1000.0 / v if v > 0 else float("inf")))
    text = "\n".join(lines)
    (logger.info if logger else print)("\n" + text)


class _WallTimer:
    def start(self):
        self.t0 = time.time() * 1000

    def end(self):
        self.t1 = time.time() * 1000

    def elapsed_time(self):
        return self.t1 - self.t0


class _EventTimer:
    def __init__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)

    def start(self):
        self.a.record()

    def end(self):
        self.b.record()

    def elapsed_time(self):
        return self.a.elapsed_time(self.b)


class timer:
    """Context manager: ``with timer('name'): ...`` appends one measurement under ``name``."""

    def __init__(self, name):
        self.t = _EventTimer() if _is_cuda else _WallTimer()
        _timer_history.setdefault(name, []).append(self.t)

    def __enter__(self):
        self.t.start()

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.t.end()
