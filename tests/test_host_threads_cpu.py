"""Host-side helpers of Estimator.train's streaming path (recsys_amd/estimator.py): the launch thread that issues staged
windows in order, the optional input thread, and the resumable iterator that keeps pulled-ahead batches between train()
calls.  Pure host logic: no GPU, no HIP calls."""
import threading
import time

import pytest

from recsys_amd.estimator import _InputThread, _LaunchThread, _Resumable


def test_launch_thread_runs_in_order_and_returns_the_last_result():
    lt = _LaunchThread("cpu")
    out = []
    for i in range(20):
        lt.submit(lambda i=i: (time.sleep(0.001), out.append(i), i * i)[-1])
    assert lt.drain() == 19 * 19
    assert out == list(range(20))
    lt.close()
    assert not lt._t.is_alive()


def test_launch_thread_surfaces_errors_at_drain_and_keeps_working():
    lt = _LaunchThread("cpu")

    def boom():
        raise ValueError("launch failed")

    lt.submit(boom)
    skipped = []
    lt.submit(lambda: skipped.append(1))           # (queued behind a failed launch: not run)
    with pytest.raises(ValueError, match="launch failed"):
        lt.drain()
    assert skipped == []
    lt.submit(lambda: 7)
    assert lt.drain() == 7
    lt.close()


def test_launch_thread_submit_is_bounded_by_one_pending_launch():
    """The training thread may stage one window ahead of the launch in progress, not more (a staging buffer is reused every
    `RSX_WINDOW_SETS` windows)."""
    lt = _LaunchThread("cpu")
    gate = threading.Event()
    lt.submit(gate.wait)                           # in progress
    lt.submit(lambda: 1)                           # pending
    t = threading.Thread(target=lambda: lt.submit(lambda: 2), daemon=True)
    t.start()
    t.join(timeout=0.2)
    assert t.is_alive()                            # the third submit waits for the queue
    gate.set()
    t.join(timeout=5)
    assert not t.is_alive() and lt.drain() == 2
    lt.close()


def test_input_thread_preserves_order_end_and_errors():
    it = _InputThread(iter(range(100)), depth=4)
    assert list(it) == list(range(100))
    with pytest.raises(StopIteration):
        next(it)                                   # stays exhausted
    it.close()

    def gen():
        yield 1
        yield 2
        raise RuntimeError("corrupt record")

    it = _InputThread(gen(), depth=4)
    assert next(it) == 1 and next(it) == 2
    with pytest.raises(RuntimeError, match="corrupt record"):
        next(it)
    it.close()


def test_input_thread_close_stops_the_producer_and_closes_the_generator():
    closed = []

    def gen():
        try:
            i = 0
            while True:
                yield i
                i += 1
        finally:
            closed.append(True)

    it = _InputThread(gen(), depth=2)
    assert next(it) == 0
    it.close()
    assert closed == [True] and not it._t.is_alive()


def test_resumable_keeps_pulled_ahead_batches_between_calls():
    """train_and_evaluate hands ONE iterator to successive Estimator.train(steps=...) calls; an input thread that pulled
    batches ahead of the first call must not lose them."""
    r = _Resumable(lambda: iter(range(50)))
    a = iter(r()).threaded(8)
    first = [next(a) for _ in range(10)]
    time.sleep(0.05)                               # (the thread runs ahead meanwhile)
    b = iter(r()).threaded(8)                      # the next train() call
    assert b is a
    rest = list(b)
    assert first + rest == list(range(50)) and r.exhausted
