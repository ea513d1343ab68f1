"""ctypes binding of libpcu_hip.so (C ABI in include/pcu_hip.h). No fallback: if the library is missing the
import of any operator fails loudly; if no GPU is visible, creating a context raises RuntimeError."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCU_HIP_LIBRARY") or os.path.join(_HERE, "libpcu_hip.so")    # override: A/B builds of the same ABI

PTRS_ON_DEVICE = 1
SQUARED = 2
NO_TIE_ORDER = 4
TIME_PHASES = 8
TIME_KERNELS = 16
STREAM_GIVEN = 32

ERR_INVALID = -1
ERR_CANCELLED = -4


class Stats(ctypes.Structure):
    _fields_ = [("n_queries", ctypes.c_int64), ("n_escalated", ctypes.c_int64), ("n_tie_flagged", ctypes.c_int64),
                ("n_tie_true", ctypes.c_int64), ("n_passes", ctypes.c_int32), ("n_grid_builds", ctypes.c_int32),
                ("ms_index", ctypes.c_float), ("ms_search", ctypes.c_float), ("ms_total", ctypes.c_float),
                ("ms_tie", ctypes.c_float), ("ms_kernel_search", ctypes.c_float),
                ("n_kernel_search", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None
_lock = threading.RLock()

CANCEL_BY_REQUEST = 1       # pcu_hip_cancel_source(): pcu_hip_cancel() ...
CANCEL_BY_SIGINT = 2        # ... or the chained SIGINT handler

_COMPUTE_ENTRY_POINTS = [f"pcu_hip_{op}_{suf}" for suf in ("f32", "f64") for op in (
    "knn", "one_sided_hausdorff", "hausdorff", "chamfer", "index_create", "index_knn", "hausdorff_batch", "chamfer_batch", "normals_knn",
    "normals_ball", "dedup", "pairwise", "sinkhorn", "dot", "debug_kd_tree")] + [
    "pcu_hip_morton_encode", "pcu_hip_morton_decode", "pcu_hip_morton_addsub", "pcu_hip_morton_knn"] + [
    f"pcu_hip_voxel_downsample_{sp}_{sa}" for sp in ("f32", "f64") for sa in ("f32", "f64")]


def _after_call(rc, func, arguments):
    """ctypes errcheck of every compute entry point. A call that SIGINT made return early is decided by the interpreter's own handler,
    as in the reference (`if (PyErr_CheckSignals() != 0) throw`): PyErr_CheckSignals runs the pending Python-level handlers on the main
    thread -- the default one raises KeyboardInterrupt, which leaves through here; if nothing was raised (a handler that only takes note, or
    a worker thread: the reference's workers see 0 too) the abandoned call is run again with the same arguments. An explicit
    pcu.cancel() always ends the call (check() raises KeyboardInterrupt)."""
    if rc == ERR_CANCELLED and _lib is not None and _lib.pcu_hip_cancel_source() == CANCEL_BY_SIGINT:
        ctypes.pythonapi.PyErr_CheckSignals()        # (PyDLL: an exception set by the handler is raised from this line)
        return func(*arguments)
    return rc


def _preload_hip_runtime():
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7)
    and look it up by the unversioned name, so if /opt/rocm's copy were loaded first a later `import torch` would
    load a second runtime and find no GPU. When torch is installed (and not yet imported) its copy is therefore
    loaded first; libpcu_hip.so's NEEDED libamdhip64.so.7 then binds to it, and so does torch later."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). point_cloud_utils_amd has no CPU fallback.")
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        L.pcu_hip_last_error.restype = ctypes.c_char_p
        L.pcu_hip_version.restype = ctypes.c_char_p
        L.pcu_hip_ctx_workspace_bytes.restype = ctypes.c_int64
        L.pcu_hip_ctx_workspace_bytes.argtypes = [ctypes.c_void_p]
        L.pcu_hip_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.pcu_hip_ctx_destroy.argtypes = [ctypes.c_void_p]
        L.pcu_hip_ctx_set_cell_occupancy.argtypes = [ctypes.c_void_p, ctypes.c_double]
        vp, i64, ci, u = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint
        for suf in ("f32", "f64"):
            getattr(L, "pcu_hip_knn_" + suf).argtypes = [vp, vp, i64, vp, i64, ci, ci, vp, vp, u, vp, vp]
            getattr(L, "pcu_hip_one_sided_hausdorff_" + suf).argtypes = [vp, vp, i64, vp, i64, ci, vp, vp, vp, u, vp, vp]
            getattr(L, "pcu_hip_hausdorff_" + suf).argtypes = [vp, vp, i64, vp, i64, ci, vp, vp, vp, u, vp, vp]
            getattr(L, "pcu_hip_chamfer_" + suf).argtypes = [vp, vp, i64, vp, i64, ctypes.c_double, ci, vp, vp, vp, u, vp, vp]
        for suf in ("f32", "f64"):
            getattr(L, "pcu_hip_debug_kd_tree_" + suf).argtypes = [vp, vp, i64, ci, vp, vp]
            getattr(L, "pcu_hip_index_create_" + suf).argtypes = [vp, vp, i64, ci, u, vp, ctypes.POINTER(ctypes.c_void_p)]
            getattr(L, "pcu_hip_index_knn_" + suf).argtypes = [vp, vp, vp, i64, ci, ci, vp, vp, u, vp, vp]
            getattr(L, "pcu_hip_hausdorff_batch_" + suf).argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, u, vp, vp]
            getattr(L, "pcu_hip_chamfer_batch_" + suf).argtypes = [vp, ci, vp, vp, vp, vp, ctypes.c_double, ci, vp, u, vp, vp]
            getattr(L, "pcu_hip_normals_knn_" + suf).argtypes = [vp, vp, i64, vp, ci, ci, ctypes.c_double, vp, vp, u, vp, vp]
            getattr(L, "pcu_hip_normals_ball_" + suf).argtypes = [vp, vp, i64, vp, ctypes.c_double, ci, ci, ci, ctypes.c_double, vp, vp, u, vp, vp]
        L.pcu_hip_morton_encode.argtypes = [vp, vp, i64, vp, u, vp]
        L.pcu_hip_morton_decode.argtypes = [vp, vp, i64, vp, u, vp]
        L.pcu_hip_morton_addsub.argtypes = [vp, vp, vp, i64, ci, vp, u, vp]
        L.pcu_hip_morton_knn.argtypes = [vp, vp, i64, vp, i64, ci, ci, vp, u, vp]
        for sp in ("f32", "f64"):
            for sa in ("f32", "f64"):
                getattr(L, f"pcu_hip_voxel_downsample_{sp}_{sa}").argtypes = [vp, vp, i64, vp, i64, ci, vp, vp, vp, ci, vp, vp, vp, u, vp]
            getattr(L, "pcu_hip_dedup_" + sp).argtypes = [vp, vp, i64, ctypes.c_double, vp, vp, vp, vp, u, vp]
        for sp in ("f32", "f64"):
            getattr(L, "pcu_hip_pairwise_" + sp).argtypes = [vp, vp, vp, i64, i64, i64, i64, ctypes.c_double, vp, u, vp]
            getattr(L, "pcu_hip_sinkhorn_" + sp).argtypes = [vp, vp, vp, vp, i64, i64, i64, ctypes.c_double, ci, ctypes.c_double, vp, vp, u, vp]
            getattr(L, "pcu_hip_dot_" + sp).argtypes = [vp, vp, vp, i64, vp, u, vp]
        L.pcu_hip_ctx_set_batch_lanes.argtypes = [vp, ci]
        L.pcu_hip_index_size.restype = ctypes.c_int64
        L.pcu_hip_index_size.argtypes = [vp]
        L.pcu_hip_index_destroy.argtypes = [vp]
        L.pcu_hip_index_destroy.restype = None
        L.pcu_hip_cancel.restype = None
        L.pcu_hip_watch_sigint.argtypes = [ci]
        L.pcu_hip_cancel_source.restype = ci
        # Ctrl-C during a long call. The reference polls PyErr_CheckSignals() per query and aborts when it returns non-zero, i.e. when the
        # interpreter's SIGINT handler RAISES (src/point_cloud_distance.cpp:60-75, 96-98): KeyboardInterrupt with the default handler; a custom
        # handler that only records the request lets the computation finish. Same here: the library chains a C handler in front of the
        # installed one (pcu_hip_watch_sigint), a call in flight returns PCU_HIP_ERR_CANCELLED early, and _after_call() below then asks the
        # interpreter (PyErr_CheckSignals): a raising handler ends the call with its exception, a non-raising one has the call run again.
        # PCU_HIP_NO_SIGINT=1 leaves the process's signal handlers alone (then a call runs to its end before the interrupt is seen).
        if os.environ.get("PCU_HIP_NO_SIGINT", "0") in ("", "0"):
            L.pcu_hip_watch_sigint(1)
        for name in _COMPUTE_ENTRY_POINTS:
            getattr(L, name).errcheck = _after_call
        _lib = L
    return _lib


def last_error():
    return lib().pcu_hip_last_error().decode("utf-8", "replace")


def device_count():
    return int(lib().pcu_hip_device_count())


def default_device():
    return int(os.environ.get("PCU_HIP_DEVICE", "0"))


_tls = threading.local()


class _ThreadCtxs:
    """The contexts one thread created. When the thread ends, its thread-local storage drops this object and the
    contexts (streams, multi-GB workspaces) are destroyed with it instead of living until the process exits."""

    def __init__(self):
        self.by_device = {}

    def __del__(self):
        try:
            import sys
            if sys.is_finalizing():          # process exit: the HIP runtime reclaims everything itself
                return
            L = _lib
            for h in self.by_device.values():
                with _lock:
                    _ctxs.discard(h.value)
                if L is not None:
                    L.pcu_hip_ctx_destroy(h)
        except Exception:
            pass


_ctxs = set()        # handles of all live contexts (diagnostics)


def ctx(device=None):
    """Per-(device, thread) context (stream + grow-only workspace), created on first use. A context is not
    thread-safe; giving every Python thread its own lets independent calls from a thread pool overlap. A thread's
    contexts are destroyed when the thread exits."""
    if device is None:
        device = default_device()
    device = int(device)
    mine = getattr(_tls, "ctxs", None)
    if mine is None:
        mine = _tls.ctxs = _ThreadCtxs()
    c = mine.by_device.get(device)
    if c is None:
        L = lib()
        h = ctypes.c_void_p()
        rc = L.pcu_hip_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"point_cloud_utils_amd: cannot create a GPU context on device {device}: "
                               f"{last_error()} (this package has no CPU fallback)")
        c = mine.by_device[device] = h
        with _lock:
            _ctxs.add(h.value)
    return c


def set_cell_occupancy(points_per_cell, device=None):
    lib().pcu_hip_ctx_set_cell_occupancy(ctx(device), float(points_per_cell))


def check(rc):
    if rc == 0:
        return
    msg = last_error()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_CANCELLED:         # pcu_hip_cancel() from another thread (a SIGINT was settled by _after_call: the interpreter's handler raised, or the call was re-run)
        raise KeyboardInterrupt(msg)
    raise RuntimeError(f"libpcu_hip: {msg}")
