"""CPU: SessionPool plumbing (ordering, exceptions through futures, clean shutdown) with stand-in sessions."""
import threading
import time

import pytest

from conftest import sub


def test_pool_orders_results_and_propagates_errors():
    made = []

    def make(i):
        made.append(i)
        return {"id": i, "lock": threading.Lock(), "calls": 0}

    def job(sess, x, delay):
        with sess["lock"]:                                  # a session is never used by two jobs at once
            sess["calls"] += 1
            time.sleep(delay)
            if x < 0:
                raise ValueError("bad batch")
            return x * x, sess["id"]

    with sub("pool").SessionPool(make, n=3) as pool:
        assert made == [0, 1, 2]
        out = pool.map(job, [(k, 0.02 * (k % 3)) for k in range(9)])
        assert [o[0] for o in out] == [k * k for k in range(9)]
        assert len({o[1] for o in out}) > 1                 # more than one session did work
        with pytest.raises(ValueError, match="bad batch"):
            pool.submit(job, -1, 0.0).result()
        assert pool.submit(job, 4, 0.0).result()[0] == 16   # the worker survives a failed job
    assert sum(s["calls"] for s in pool.sessions) == 11
    with pytest.raises(ValueError):
        sub("pool").SessionPool(make, n=0)
