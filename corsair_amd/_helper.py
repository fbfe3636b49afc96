"""One helper thread + HIP stream per calling thread, for launch sequences that are independent of what
the caller enqueues next (kernel maps of the deeper levels while the first convolutions run; the vanilla
RANSACs while the symmetry stages run).  The library's scratch cache is per host thread and ordered by
that thread's stream, so the helper needs no locking; results cross over through HIP events."""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor

import torch

_tls = threading.local()


def pool():
    if getattr(_tls, "pool", None) is None:
        _tls.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="corsair-helper")
        _tls.streams = {}
    return _tls.pool


def stream(dev):
    pool()
    key = (dev.type, dev.index)
    if key not in _tls.streams:
        _tls.streams[key] = torch.cuda.Stream(device=dev)
    return _tls.streams[key]


def submit(dev, fn, after=None):
    """Run fn() on the helper thread with the helper stream current; the stream first waits for the
    event `after` (default: everything the caller's stream holds now).  Returns the future."""
    side = stream(dev)
    ready = after if after is not None else torch.cuda.current_stream(dev).record_event()

    def run():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            return fn()

    return pool().submit(run)
