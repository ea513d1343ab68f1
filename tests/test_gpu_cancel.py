"""Interruptibility (-m gpu). The reference polls PyErr_CheckSignals() per query and per kd-tree node and raises KeyboardInterrupt
(/root/reference/src/point_cloud_distance.cpp:60-75, 96-98; external/nanoflann/nanoflann.hpp:1004). Here every host-side wait of a call is a
bounded poll on a cancellation flag (csrc/pcu_hip.hip: wait_stream, wait_result_block), set by pcu.cancel() or by the SIGINT handler the
library chains in front of Python's."""
import os
import signal
import threading
import time

import numpy as np
import pytest

import oracle
from conftest import cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcu():
    import point_cloud_utils_amd as m
    from point_cloud_utils_amd import _lib
    assert _lib.device_count() > 0, "no GPU visible: the gfx950 path has no CPU fallback"
    return m


def _still_correct(pcu, kind):
    """The context works after an abandoned call: a row-based call, a fused call (the "no memset" fill words of the one-pass build) and a
    tie-order call, against the oracle."""
    q, r = cloud(71, 150_000, np.float32), cloud(72, 140_000, np.float32)
    d, c = pcu.k_nearest_neighbors(q, r, 2)
    d0, c0 = oracle.k_nearest_neighbors(q, r, 2, kind=kind)
    assert np.array_equal(c, c0) and np.array_equal(d, d0)
    for _ in range(3):          # (both parities of the fill words)
        assert abs(float(pcu.chamfer_distance(q, r)) - float(oracle.chamfer_distance(q, r, kind=kind))) <= 1e-4 * float(pcu.chamfer_distance(q, r))
    assert pcu.hausdorff_distance(q, r, return_index=True) == oracle.hausdorff_distance(q, r, return_index=True, kind=kind)
    dup = np.repeat(cloud(73, 3000, np.float64), 3, axis=0)
    d, c = pcu.k_nearest_neighbors(dup, dup, 4)
    d0, c0 = oracle.k_nearest_neighbors(dup, dup, 4, kind=kind)
    assert np.array_equal(c, c0) and np.array_equal(d, d0)


def _long_call(pcu, started):
    """A call that takes tens of milliseconds at least (two 20M-point float32 clouds from host memory: 240 MB of H2D copies each, an index
    build and the search), repeated until something interrupts it."""
    q, r = cloud(81, 20_000_000, np.float32), cloud(82, 20_000_000, np.float32)
    t0 = time.perf_counter()
    started.set()
    for _ in range(400):            # (minutes if nothing stops it)
        pcu.k_nearest_neighbors(q, r, 1)
    return time.perf_counter() - t0


def test_cancel_from_another_thread(pcu, oracle_kind):
    started = threading.Event()
    def canceller():
        started.wait()
        for _ in range(200):        # keep asking until the main thread has left the loop (a request made between two calls is dropped)
            time.sleep(0.02)
            pcu.cancel()
            if done.is_set(): break
    done = threading.Event()
    th = threading.Thread(target=canceller); th.start()
    t0 = time.perf_counter()
    try:
        with pytest.raises(KeyboardInterrupt):
            _long_call(pcu, started)
    finally:
        done.set(); th.join()
    assert time.perf_counter() - t0 < 60.0
    _still_correct(pcu, oracle_kind)


def test_sigint_raises_keyboard_interrupt(pcu, oracle_kind):
    if os.environ.get("PCU_HIP_NO_SIGINT", "0") not in ("", "0"):
        pytest.skip("PCU_HIP_NO_SIGINT: the library leaves the signal handlers alone")
    assert threading.current_thread() is threading.main_thread()
    started, left = threading.Event(), threading.Event()
    def killer():
        started.wait()
        if left.wait(0.05):           # the main thread has already left the guarded block (some other failure): do NOT interrupt pytest itself
            return
        os.kill(os.getpid(), signal.SIGINT)
    th = threading.Thread(target=killer); th.start()
    t0 = time.perf_counter()
    try:
        with pytest.raises(KeyboardInterrupt):
            try:
                _long_call(pcu, started)
            finally:
                left.set()            # (set before the block is left, whatever leaves it; a SIGINT already on its way still lands inside pytest.raises)
                started.set()
    finally:
        th.join()
    assert time.perf_counter() - t0 < 60.0            # 400 calls would take minutes
    _still_correct(pcu, oracle_kind)
