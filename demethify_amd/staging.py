"""Host-side overlap for the restart loop (demethify/demethify.py:165-171,195-201 draws a fresh initialisation per
restart and solves it, one after the other): the initialisation of restart k + 1 -- legacy-numpy draws, NNLS,
page-locked copy, upload on a stream of its own -- runs in a worker thread while the GPU iterates restart k.

Plumbing only: numpy's generators and the library's copies release the GIL.  No torch here: the page-locked buffers
come from dmf_host_alloc, the upload from dmf_stage_upload (a copy stream of the context's own).
"""
from __future__ import annotations

import threading

import numpy as np


class Prefetcher:
    """Runs ``fn(item)`` for every item of ``items`` on ``workers`` threads, at most ``depth`` results ahead of the
    consumer, and yields (item, result) in the order of ``items``; an exception raised by ``fn`` is re-raised in the
    consumer at the position of its item."""

    def __init__(self, items, fn, depth: int = 1, workers: int = 1):
        from concurrent.futures import ThreadPoolExecutor

        self._items, self._fn = list(items), fn
        self._depth = max(1, int(depth))
        self._pool = ThreadPoolExecutor(max_workers=max(1, min(int(workers), self._depth)),
                                        thread_name_prefix="dmf-prefetch")
        self._pending: list = []
        self._next = 0
        self._closed = False
        self._fill()

    def _fill(self):
        while not self._closed and self._next < len(self._items) and len(self._pending) < self._depth:
            item = self._items[self._next]
            self._pending.append((item, self._pool.submit(self._fn, item)))
            self._next += 1

    def __iter__(self):
        while self._pending:
            item, fut = self._pending.pop(0)
            try:
                result = fut.result()
            except BaseException:
                self.close()
                raise
            self._fill()
            yield item, result
        self.close()

    def close(self):
        """Stop early (the consumer left its loop): nothing further is started, the worker threads end."""
        if not self._closed:
            self._closed = True
            for _, fut in self._pending:
                fut.cancel()
            self._pending = []
            self._pool.shutdown(wait=True)

    def is_alive(self):
        return not self._closed

    def join(self, timeout=None):
        self.close()


class DeviceArray:
    """A float64 array in HBM that the library uploaded (dmf_stage_upload): shape + device pointer, released with the
    object.  ``Solver`` takes a pair of them in place of host arrays."""

    is_cuda = True

    def __init__(self, ctx, ptr, shape):
        import weakref

        self.ctx, self._ptr, self.shape = ctx, ptr, tuple(int(x) for x in shape)
        self._fin = weakref.finalize(self, _release, ctx, ptr)
        # every value inside [0, 1]?  (to_device looks at small arrays -- the proportions -- on the uploading thread, so
        # that dmf_solver_create need not bring alpha0 back to the host for it: DMF_INIT_IN_UNIT_RANGE)
        self.in_unit_range = None

    def data_ptr(self) -> int:
        return self._ptr

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
        n = int(np.prod(self.shape))
        shape = tuple(n // -int(np.prod(shape)) if x == -1 else int(x) for x in shape)
        if int(np.prod(shape)) != n:
            raise ValueError(f"cannot reshape {self.shape} to {shape}")
        view = DeviceArray.__new__(DeviceArray)
        view.ctx, view._ptr, view.shape, view._fin, view._base = self.ctx, self._ptr, shape, None, self
        view.in_unit_range = self.in_unit_range
        return view

    def close(self):
        if self._fin is not None:
            self._fin()


def _release(ctx, ptr):
    if getattr(ctx, "_h", None):  # (a context that is already closed took its pool with it)
        ctx._lib.dmf_stage_free(ctx._h, ptr)


def to_device(arrays, ctx):
    """Upload float64 host arrays to the context's GPU: copy into page-locked buffers (pooled per shape -- locking
    32 MB of pages costs far more than copying them), then dmf_stage_upload on the context's copy stream.  Returns
    DeviceArrays that are complete when this returns.  Meant for a worker thread: the copy engine works beside the
    solver's kernels, and every call here releases the GIL."""
    import ctypes as C

    from . import _lib as L

    out = []
    for a in arrays:
        a = np.asarray(a, dtype=np.float64)
        buf = _borrow(a.shape)
        try:
            np.copyto(buf, a)
            dev = C.c_void_p()
            L.check(ctx._lib.dmf_stage_upload(ctx._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(dev)),
                    "dmf_stage_upload")
        finally:
            _give_back(buf)
        out.append(DeviceArray(ctx, dev.value, a.shape))
        if a.size <= (1 << 20):
            out[-1].in_unit_range = bool(a.size == 0 or (a.min() >= 0.0 and a.max() <= 1.0))
    return out


def indices_to_device(idx, ctx):
    """Upload an int64 index array (a bootstrap replicate's row draw) the same way: page-locked copy, dmf_stage_upload.
    The DeviceArray's bytes are the int64 values (Problem.gather takes it in place of the host array)."""
    import ctypes as C

    from . import _lib as L

    idx = np.ascontiguousarray(idx, dtype=np.int64)
    buf = _borrow(idx.shape)  # (float64 staging buffers: the same eight bytes per element)
    try:
        np.copyto(buf.view(np.int64), idx)
        dev = C.c_void_p()
        L.check(ctx._lib.dmf_stage_upload(ctx._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(dev)),
                "dmf_stage_upload")
    finally:
        _give_back(buf)
    return DeviceArray(ctx, dev.value, idx.shape)


def reserve(shapes, count: int = 1):
    """Page-lock ``count`` staging buffers per shape now (a restart job does this once, before its first restart,
    instead of inside the first uploads)."""
    held = [[_borrow(tuple(shape)) for _ in range(count)] for shape in shapes]
    for group in held:
        for buf in group:
            _give_back(buf)


_free: dict = {}
_free_lock = threading.Lock()


def _borrow(shape):
    from .tables import _host_matrix

    with _free_lock:
        bufs = _free.get(tuple(shape))
        if bufs:
            return bufs.pop()
    return _host_matrix(tuple(shape), np.float64)


def _give_back(buf):
    with _free_lock:
        _free.setdefault(tuple(buf.shape), []).append(buf)
