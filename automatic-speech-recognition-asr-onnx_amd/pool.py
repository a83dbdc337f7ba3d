"""Several batches in flight on one GPU.

A native session owns its HIP stream and every per-batch buffer; the weight arena can be borrowed (`arena_device_ptr`), so N sessions
of one model cost N workspaces and one copy of the weights. `SessionPool` keeps N sessions and N host threads (the C ABI runs outside
the GIL under ctypes): the compute-bound encoder / prefill of one batch overlaps the latency-bound decode chain of another. Measured
per MI355X: Whisper-large-v3 +59 % (8 s) / +53 % (30 s), Qwen3-ASR-0.6B +50 % at three sessions, SenseVoiceSmall
+15 % at two. The reference has no counterpart (one onnxruntime session, one utterance at a time); results per batch are exactly those
of a lone session.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import Future
from typing import Callable, Iterable, Sequence


class SessionPool:
    def __init__(self, make_session: Callable[[int], object], n: int = 2, device_id: int | None = None):
        """make_session(i) -> the i-th session (create them with a shared, device-resident arena). device_id: bind the worker threads
        to this device through torch, when torch is in use by the caller (optional)."""
        if n < 1:
            raise ValueError("SessionPool needs at least one session")
        self.sessions = [make_session(i) for i in range(n)]
        self._jobs: queue.Queue = queue.Queue()
        self._device_id = device_id
        self._threads = [threading.Thread(target=self._work, args=(s,), daemon=True) for s in self.sessions]
        for t in self._threads:
            t.start()

    def _work(self, session):
        if self._device_id is not None:
            import torch
            torch.cuda.set_device(self._device_id)
        while True:
            job = self._jobs.get()
            if job is None:
                return
            fn, args, fut = job
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                fut.set_result(fn(session, *args))
            except BaseException as e:                     # delivered to the caller through the future
                fut.set_exception(e)

    def submit(self, fn: Callable, *args) -> Future:
        """Queue fn(session, *args) for the next free session."""
        fut: Future = Future()
        self._jobs.put((fn, args, fut))
        return fut

    def map(self, fn: Callable, batches: Iterable[Sequence]) -> list:
        """fn(session, *batch) for every batch, results in submission order."""
        return [f.result() for f in [self.submit(fn, *b) for b in batches]]

    def close(self):
        for _ in self._threads:
            self._jobs.put(None)
        for t in self._threads:
            t.join()
        self._threads = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
