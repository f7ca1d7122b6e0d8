"""Micro-batcher for single-query callers (the Python twin of MicroBatcher in csrc/host/hip_index.h).

The product drives the index boundary with ONE query per call from many threads (index.cc:617); one batched call answers
1024 queries in the time ~25 single-query calls take.  Concurrent callers with the same topk join an open batch; the first
one in leads it: it keeps the batch open while an earlier batch is still searching (at most until it is full or
`window_us` has passed; `linger_us` even on an idle index), runs ONE batched search through the C ABI and hands every
caller its own result rows.  Callers with a filter / threshold / fetch_vector must not use it (their searches are not
interchangeable)."""
import threading
import time

import numpy as np


class MicroBatcher:
    def __init__(self, run, dim, np_dtype, max_batch=1024, window_us=2000, linger_us=0):
        """run(queries [n][dim], topk) -> (keys [n][topk], scores [n][topk], counts [n])"""
        self._run, self._dim, self._dt = run, dim, np_dtype
        self._max, self._window, self._linger = max(1, int(max_batch)), window_us * 1e-6, min(linger_us, window_us) * 1e-6
        self._cv = threading.Condition()
        self._open = None
        self._inflight = 0

    def search(self, query, topk):
        q = np.ascontiguousarray(query, self._dt).reshape(self._dim)
        with self._cv:
            while self._open is not None and (self._open["topk"] != topk or len(self._open["q"]) >= self._max):
                self._cv.wait()
            b, leader = self._open, False
            if b is None:
                now = time.perf_counter()
                b = {"topk": topk, "q": [], "done": False, "err": None, "deadline": now + self._window, "linger": now + self._linger}
                self._open, leader = b, True
            slot = len(b["q"])
            b["q"].append(q)
            if len(b["q"]) >= self._max:
                self._cv.notify_all()
            if leader:
                while len(b["q"]) < self._max:
                    until = b["deadline"] if self._inflight > 0 else b["linger"]
                    left = until - time.perf_counter()
                    if left <= 0:
                        break
                    self._cv.wait(left)
                self._open = None
                self._inflight += 1
                self._cv.notify_all()
        if leader:
            try:
                b["res"] = self._run(np.stack(b["q"]), topk)
            except Exception as e:   # noqa: BLE001 - handed to every caller of the batch
                b["err"] = e
            with self._cv:
                self._inflight -= 1
                b["done"] = True
                self._cv.notify_all()
        else:
            with self._cv:
                while not b["done"]:
                    self._cv.wait()
        if b["err"] is not None:
            raise b["err"]
        keys, scores, counts = b["res"]
        c = int(counts[slot])
        return keys[slot, :c], scores[slot, :c]
