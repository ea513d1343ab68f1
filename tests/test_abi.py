"""CPU tests: the C-ABI library loads and exports every symbol include/pcu_hip.h declares; host-side
validation mirrors the reference's ValueErrors; no compute is attempted without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pcu_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pcu_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from point_cloud_utils_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"libpcu_hip.so does not export {n}"
    assert b"gfx950" in L.pcu_hip_version()


def test_only_the_c_abi_is_exported():
    """The dynamic symbol table holds the entry points of include/pcu_hip.h and nothing else (csrc/export.map): no kernel stubs, no C++
    template instantiations, no torch / Python symbols -- a C-ABI boundary."""
    import shutil
    import subprocess
    from point_cloud_utils_amd import _lib
    if not shutil.which("nm"):
        pytest.skip("nm not in this image")
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.split()[-2:-1] and ln.split()[-2] in ("T", "D", "B", "W", "V"))
    assert exported == _declared(), sorted(set(exported) ^ set(_declared()))


def test_stats_struct_size_matches_header():
    from point_cloud_utils_amd import _lib
    assert ctypes.sizeof(_lib.Stats) == 4 * 8 + 2 * 4 + 4 * 4 + 4 + 4


def test_validation_errors_match_reference_texts():
    import point_cloud_utils_amd as pcu
    a = np.random.rand(10, 3); b = np.random.rand(5, 3)
    with pytest.raises(ValueError, match=r"Invalid value for k \(0\) must be greater than 0\."):
        pcu.k_nearest_neighbors(a, b, 0)
    with pytest.raises(ValueError, match="Invalid input set with zero elements: query_points and dataset_points"):
        pcu.k_nearest_neighbors(np.zeros((0, 3)), b, 1)
    with pytest.raises(ValueError, match=r"Only 3D inputs are supported.*dataset_points.shape = \(5, 2\)"):
        pcu.k_nearest_neighbors(a, np.random.rand(5, 2), 1)
    with pytest.raises(ValueError, match="Invalid input set with zero elements: source and targets"):
        pcu.one_sided_hausdorff_distance(a, np.zeros((0, 3)))
    with pytest.raises(ValueError, match="Only 3D inputs are supported: source and targets"):
        pcu.hausdorff_distance(np.random.rand(4, 4), b)
    with pytest.raises(ValueError, match="Invalid scalar type"):
        pcu.k_nearest_neighbors(a.astype(np.float32), b, 1)          # dtype mismatch (npe_matches)
    with pytest.raises(ValueError, match="Invalid scalar type"):
        pcu.chamfer_distance(a.astype(np.int32), b.astype(np.int32))


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must raise, not compute on the CPU."""
    import point_cloud_utils_amd as pcu
    from point_cloud_utils_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pcu.k_nearest_neighbors(np.random.rand(10, 3), np.random.rand(5, 3), 1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "point_cloud_utils_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libpcu_oracle" not in txt and "libpcu_ref" not in txt, f


def test_dataset_index_validates_before_touching_the_gpu():
    import point_cloud_utils_amd as pcu
    with pytest.raises(ValueError, match="Invalid scalar type"):
        pcu.DatasetIndex(np.zeros((5, 3), dtype=np.int32))
    with pytest.raises(ValueError, match=r"shape \(m, 3\)"):
        pcu.DatasetIndex(np.zeros((5, 2)))
    with pytest.raises(ValueError, match=r"shape \(m, 3\)"):
        pcu.DatasetIndex(np.zeros((0, 3)))


def test_timing_switch_is_a_module_level_setting():
    import point_cloud_utils_amd as pcu
    old = pcu.set_timing(2)
    try:
        assert pcu.set_timing(0) == 2
        from point_cloud_utils_amd import _lib
        pcu.set_timing(1)
        assert pcu._flags() & _lib.TIME_KERNELS and not (pcu._flags() & _lib.TIME_PHASES)
    finally:
        pcu.set_timing(old)


def test_sigint_handler_chains_to_the_interpreters():
    """pcu_hip_watch_sigint (installed when the library is loaded, _lib.py) sits IN FRONT of Python's SIGINT handler and calls it: Ctrl-C must
    still raise KeyboardInterrupt in the interpreter -- with the watch on, after switching it off (the previous handler is restored) and on
    again. The reference's counterpart is its PyErr_CheckSignals() polling (/root/reference/src/point_cloud_distance.cpp:60-75, 96-98). No GPU
    needed; the cancellation of a call in flight is tests/test_gpu_cancel.py."""
    import signal
    import subprocess
    import sys
    import time
    code = (
        "import os, signal, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from point_cloud_utils_amd import _lib\n"
        "L = _lib.lib()\n"
        "def hit():\n"
        "    try:\n"
        "        os.kill(os.getpid(), signal.SIGINT); time.sleep(2.0)\n"
        "    except KeyboardInterrupt:\n"
        "        return True\n"
        "    return False\n"
        "ok = [hit()]\n"
        "assert L.pcu_hip_watch_sigint(0) == 0; ok.append(hit())\n"
        "assert L.pcu_hip_watch_sigint(1) == 0; ok.append(hit())\n"
        "L.pcu_hip_cancel()\n"
        "# a host that ignores SIGINT when the watch is installed keeps ignoring it\n"
        "assert L.pcu_hip_watch_sigint(0) == 0\n"
        "signal.signal(signal.SIGINT, signal.SIG_IGN); assert L.pcu_hip_watch_sigint(1) == 0\n"
        "os.kill(os.getpid(), signal.SIGINT); time.sleep(0.2); ok.append('ignored')\n"
        "print(ok)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("[True, True, True, 'ignored']"), (r.stdout, r.stderr[-2000:])


def test_sigint_rule_is_the_references():
    """The reference abandons a call only if the interpreter's SIGINT handler RAISES (`if (PyErr_CheckSignals() != 0) throw`,
    /root/reference/src/point_cloud_distance.cpp:60-75): (i) the library records who asked for the cancellation; (ii) a handler installed with
    signal.signal() AFTER the library's (which replaces it) is chained again by the next compute call; (iii) _lib._after_call -- the errcheck of
    every compute entry point -- re-issues a SIGINT-abandoned call when the handler only took note, and lets a raising handler's exception
    through; (iv) pcu.cancel() always ends the call. No GPU needed (the compute call used for (ii) fails on its null context, after its guard)."""
    import subprocess
    import sys
    code = (
        "import ctypes, os, signal, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from point_cloud_utils_amd import _lib\n"
        "L = _lib.lib()\n"
        "L.pcu_hip_cancel(); assert L.pcu_hip_cancel_source() == _lib.CANCEL_BY_REQUEST\n"
        "noted = []\n"
        "signal.signal(signal.SIGINT, lambda s, f: noted.append(s))          # replaces the library's handler\n"
        "os.kill(os.getpid(), signal.SIGINT); time.sleep(0.05)\n"
        "assert noted == [signal.SIGINT] and L.pcu_hip_cancel_source() == _lib.CANCEL_BY_REQUEST      # (ours is gone: nothing recorded)\n"
        "rc = L.pcu_hip_knn_f32(None, None, 0, None, 0, 1, 10, None, None, 0, None, None)\n"
        "assert rc == _lib.ERR_INVALID, rc                                    # null context; the guard has re-armed the chain\n"
        "os.kill(os.getpid(), signal.SIGINT); time.sleep(0.05)\n"
        "assert noted == [signal.SIGINT] * 2 and L.pcu_hip_cancel_source() == _lib.CANCEL_BY_SIGINT\n"
        "calls = []\n"
        "def fake(*a):\n"
        "    calls.append(a); return 0\n"
        "os.kill(os.getpid(), signal.SIGINT)\n"
        "assert _lib._after_call(_lib.ERR_CANCELLED, fake, (1, 2)) == 0 and calls == [(1, 2)]     # a handler that takes note: the call runs again\n"
        "assert _lib._after_call(_lib.ERR_INVALID, fake, (3,)) == _lib.ERR_INVALID and len(calls) == 1\n"
        "signal.signal(signal.SIGINT, signal.default_int_handler)\n"
        "L.pcu_hip_knn_f32(None, None, 0, None, 0, 1, 10, None, None, 0, None, None)\n"
        "try:\n"
        "    signal.pthread_sigmask(signal.SIG_BLOCK, [signal.SIGINT]); os.kill(os.getpid(), signal.SIGINT)\n"
        "    signal.pthread_sigmask(signal.SIG_UNBLOCK, [signal.SIGINT])\n"
        "    _lib._after_call(_lib.ERR_CANCELLED, fake, (4,)); time.sleep(1.0); raised = False\n"
        "except KeyboardInterrupt:\n"
        "    raised = True\n"
        "assert raised and len(calls) == 1\n"
        "L.pcu_hip_cancel()\n"
        "assert _lib._after_call(_lib.ERR_CANCELLED, fake, (5,)) == _lib.ERR_CANCELLED and len(calls) == 1   # pcu.cancel(): never re-issued\n"
        "try:\n"
        "    _lib.check(_lib.ERR_CANCELLED); ok = False\n"
        "except KeyboardInterrupt:\n"
        "    ok = True\n"
        "assert ok\n"
        "print('rule ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("rule ok"), (r.stdout, r.stderr[-3000:])
