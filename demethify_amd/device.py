"""Object wrappers over the C-ABI handles: Context (one per GPU), Problem (V, D, R_trunc resident
in HBM) and Solver (the outer loop's device-resident state).

Inputs may be host numpy arrays (uploaded by the library) or PyTorch-ROCm tensors already on
the context's GPU (borrowed through ``data_ptr()``; torch is used for nothing else).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L

_contexts: dict[int, "Context"] = {}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _host_f64(x, name):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if not np.all(np.isfinite(a)):
        # upstream lets NaN propagate silently; a device solve cannot stop on it
        raise ValueError(f"{name} holds non-finite values (use --fillna)")
    return a


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a) or hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """dmf_context: a GPU, a HIP stream and the kernel-family clocks."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._lib = L.load()
        h = C.c_void_p()
        L.check(self._lib.dmf_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)),
                "dmf_context_create")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dmf_context_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - interpreter teardown order
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        L.check(self._lib.dmf_context_synchronize(self._h), "dmf_context_synchronize")

    def set_profiling(self, enabled, families=None):
        """Record HIP events around kernel launches: all families, or only ``families`` (indices into
        ``_lib.KERNEL_FAMILIES``) — every timed launch costs two event records on the stream."""
        mode = int(bool(enabled))
        if mode and families is not None:
            mode = sum(1 << (1 + int(f)) for f in families)
        L.check(self._lib.dmf_context_set_profiling(self._h, mode), "dmf_context_set_profiling")

    def set_generic(self, level: int):
        """Kernel selection for tests: 0 fastest (u16-count row pass + integer-MFMA Gram, falling back to level 4,
        then 3), 1 any-shape Gram-form kernels, 2 schedule-faithful one-launch-per-inner-step kernels, 3 the unfused
        MFMA row pass + one-pass Gram pair, 4 the first-generation fused FP64 row pass.  Set before creating Problems."""
        L.check(self._lib.dmf_context_set_generic(self._h, int(level)), "dmf_context_set_generic")

    def set_stop_confirmation(self, mode: int):
        """How step() decides |cf - cf_0| < tol: 0 (default) Gram-form cost, confirmed on the streaming cost where the
        Gram form's error bound reaches tol / 20; 1 always on streaming costs near the threshold; 2 Gram form only."""
        L.check(self._lib.dmf_context_set_stop_confirmation(self._h, int(mode)), "dmf_context_set_stop_confirmation")

    def reset_kernel_time(self):
        L.check(self._lib.dmf_context_reset_kernel_time(self._h), "dmf_context_reset_kernel_time")

    def kernel_time(self, family: int):
        """(total milliseconds, launches) accumulated for one kernel family while profiling."""
        ms, n = C.c_double(), C.c_int64()
        L.check(self._lib.dmf_context_kernel_time(self._h, int(family), C.byref(ms), C.byref(n)),
                "dmf_context_kernel_time")
        return ms.value, n.value

    # ---- single-function entry points -------------------------------------------------
    def project_simplex(self, X, z=1.0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        out = np.empty_like(X)
        K, S = X.shape
        L.check(self._lib.dmf_project_simplex(self._h, _ptr(X), K, S, float(z), 0, _ptr(out)),
                "dmf_project_simplex")
        return out

    def percentile_axis0(self, x, q):
        """``np.percentile(x, q, axis=0)`` (numpy's default "linear" method, bootstrap.py:51-54 / :75-78) on
        the device.  ``x``: (n replicates, ...) float64, a host array or a CUDA torch tensor; ``q``: a sequence
        of percentiles.  Returns an array / tensor of shape (len(q),) + x.shape[1:] of the same kind as ``x``."""
        qs = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
        if _is_torch(x) and x.is_cuda:
            import torch

            if x.dtype != torch.float64:
                raise TypeError("percentile_axis0 takes float64 data")
            x = x.contiguous()
            n, tail = x.shape[0], tuple(x.shape[1:])
            m = int(np.prod(tail, dtype=np.int64))
            out = torch.empty((len(qs),) + tail, dtype=torch.float64, device=x.device)
            torch.cuda.current_stream(x.device).synchronize()  # x was produced on torch's stream, not ours
            L.check(self._lib.dmf_percentile_axis0(self._h, _ptr(x), n, m, _ptr(qs), len(qs), L.DMF_PTR_DEVICE,
                                                   _ptr(out)), "dmf_percentile_axis0")
            return out
        if _is_torch(x):
            x = x.numpy()
        x = np.ascontiguousarray(x, dtype=np.float64)
        n, tail = x.shape[0], tuple(x.shape[1:])
        m = int(np.prod(tail, dtype=np.int64))
        out = np.empty((len(qs),) + tail, dtype=np.float64)
        L.check(self._lib.dmf_percentile_axis0(self._h, _ptr(x), n, m, _ptr(qs), len(qs), 0, _ptr(out)),
                "dmf_percentile_axis0")
        return out


def get_context(device: int | None = None) -> Context:
    """Process-wide context cache; default device = LOCAL_RANK (one process per GPU) or 0."""
    if device is None:
        # DEMETHIFY_DEVICE overrides the one-process-per-GPU default (rehearsing N ranks on one GPU)
        device = int(os.environ.get("DEMETHIFY_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx


class Problem:
    """dmf_problem: meth_frequency (N x S), counts (N x S), R_trunc (N x n_c or None)."""

    def __init__(self, ctx: Context, V, counts, Rt=None):
        self.ctx = ctx
        self._lib = ctx._lib
        flags = 0
        if _is_torch(V):
            import torch

            flags |= L.DMF_PTR_DEVICE
            for t in (V, counts) + ((Rt,) if Rt is not None else ()):
                if not (t.is_cuda and t.is_contiguous() and t.device.index == ctx.device):
                    raise ValueError("device tensors must be contiguous and on the context's GPU")
            if V.dtype != torch.float64 or (Rt is not None and Rt.dtype != torch.float64):
                raise ValueError("V and R_trunc must be float64")
            if counts.dtype == torch.float64:
                flags |= L.DMF_COUNTS_F64
            elif counts.dtype != torch.int64:
                raise ValueError("counts must be int64 or float64")
            # the tensors were produced on torch's stream, the library reads them on its own
            torch.cuda.current_stream(V.device).synchronize()
            N, S = V.shape
        else:
            V = _host_f64(V, "meth_frequency")
            counts = np.asarray(counts)
            if counts.dtype.kind in "iub":
                counts = np.ascontiguousarray(counts, dtype=np.int64)
            else:
                counts = _host_f64(counts, "counts")
                flags |= L.DMF_COUNTS_F64
            if Rt is not None:
                Rt = _host_f64(Rt, "R_trunc")
            N, S = V.shape
        if tuple(counts.shape) != (N, S):
            raise ValueError(f"counts shape {tuple(counts.shape)} != meth_frequency shape {(N, S)}")
        n_c = 0
        if Rt is not None:
            if Rt.ndim != 2 or Rt.shape[0] != N:
                raise ValueError(f"R_trunc shape {tuple(Rt.shape)} does not match {N} CpG rows")
            n_c = int(Rt.shape[1])
        self._keep = (V, counts, Rt)  # device tensors are borrowed: keep them alive
        self.N, self.S, self.n_c = int(N), int(S), n_c
        h = C.c_void_p()
        L.check(self._lib.dmf_problem_create(ctx._h, self.N, self.S, self.n_c, _ptr(V), _ptr(counts),
                                             _ptr(Rt) if n_c else None, flags, C.byref(h)),
                "dmf_problem_create")
        self._h = h

    @classmethod
    def _from_handle(cls, ctx, h, N, S, n_c):
        self = cls.__new__(cls)
        self.ctx, self._lib, self._h = ctx, ctx._lib, h
        self.N, self.S, self.n_c = N, S, n_c
        self._keep = ()
        return self

    def gather(self, idx) -> "Problem":
        """Row-resampled copy (one bootstrap replicate, bootstrap.py:28).  idx: a host integer array, or the int64 indices
        already in HBM (staging.indices_to_device: uploaded by the thread that drew them)."""
        h = C.c_void_p()
        if getattr(idx, "is_cuda", False):
            n = int(np.prod(idx.shape))
            L.check(self._lib.dmf_problem_gather_device(self.ctx._h, self._h, C.c_void_p(idx.data_ptr()), n, C.byref(h)),
                    "dmf_problem_gather_device")
            return Problem._from_handle(self.ctx, h, n, self.S, self.n_c)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        L.check(self._lib.dmf_problem_gather(self.ctx._h, self._h, _ptr(idx), idx.size, C.byref(h)),
                "dmf_problem_gather")
        return Problem._from_handle(self.ctx, h, int(idx.size), self.S, self.n_c)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dmf_problem_destroy(self._h)
            self._h = None
            self._keep = ()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- single-function entry points ---------------------------------------------------
    def cost(self, u, alpha) -> float:
        """cost_f_w with R = [R_trunc | u] (deconvolution.py:15-17)."""
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        n_u = 0
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.N, -1)
            n_u = u.shape[1]
        if alpha.shape != (self.n_c + n_u, self.S):
            raise ValueError(f"alpha shape {alpha.shape} != {(self.n_c + n_u, self.S)}")
        out = C.c_double()
        L.check(self._lib.dmf_cost(self.ctx._h, self._h, _ptr(u) if n_u else None, n_u, _ptr(alpha), 0,
                                   C.byref(out)), "dmf_cost")
        return out.value

    def update_u(self, u, u_prev, alpha, n_iter2, a1, l_w_prev, l_w, mode=L.DMF_MODE_PARTIAL):
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.N, -1)
        u_prev = np.ascontiguousarray(u_prev, dtype=np.float64).reshape(u.shape)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        n_u = u.shape[1]
        if alpha.shape != (self.n_c + n_u, self.S):
            raise ValueError(f"alpha shape {alpha.shape} != {(self.n_c + n_u, self.S)}")
        sc = (C.c_double * 3)(float(a1), float(l_w_prev), float(l_w))
        out_u, out_up = np.empty_like(u), np.empty_like(u)
        L.check(self._lib.dmf_update_u(self.ctx._h, self._h, _ptr(u), _ptr(u_prev), _ptr(alpha), n_u,
                                       int(n_iter2), int(mode), 0, sc, _ptr(out_u), _ptr(out_up)),
                "dmf_update_u")
        return out_u, out_up, sc[0], sc[1]

    def update_alpha(self, u, alpha, alpha_prev, n_iter2, a2, l_h_prev, l_h):
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.N, -1)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        alpha_prev = np.ascontiguousarray(alpha_prev, dtype=np.float64)
        n_u = u.shape[1]
        if alpha.shape != (self.n_c + n_u, self.S) or alpha_prev.shape != alpha.shape:
            raise ValueError(f"alpha shape {alpha.shape} != {(self.n_c + n_u, self.S)}")
        sc = (C.c_double * 3)(float(a2), float(l_h_prev), float(l_h))
        out_a, out_ap = np.empty_like(alpha), np.empty_like(alpha)
        L.check(self._lib.dmf_update_alpha(self.ctx._h, self._h, _ptr(u), n_u, _ptr(alpha), _ptr(alpha_prev),
                                           int(n_iter2), 0, sc, _ptr(out_a), _ptr(out_ap)),
                "dmf_update_alpha")
        return out_a, out_ap, sc[0], sc[1]


class Solver:
    """dmf_solver: state init (deconvolution.py:192-204) + stepping of the outer loop."""

    def __init__(self, problem: Problem, u0, alpha0, mode=L.DMF_MODE_PARTIAL):
        """u0 (N x n_u) and alpha0 (K x S): host arrays, or BOTH float64 device arrays on the context's GPU
        (staging.to_device: the initialisation of the next restart, uploaded while this one iterates; or CUDA tensors)."""
        self.problem = problem
        self._lib = problem._lib
        flags = 0
        if getattr(u0, "is_cuda", False) or getattr(alpha0, "is_cuda", False):
            from .staging import DeviceArray

            for t in (u0, alpha0):
                if isinstance(t, DeviceArray):
                    ok = t.ctx is problem.ctx
                else:
                    ok = (_is_torch(t) and t.is_cuda and t.is_contiguous() and t.element_size() == 8
                          and t.is_floating_point() and t.device.index == problem.ctx.device)
                if not ok:
                    raise ValueError("u0 and alpha0 must both be host arrays, or both be float64 device arrays "
                                     "(staging.DeviceArray / contiguous CUDA tensors) on the context's GPU")
            u0 = u0.reshape(problem.N, -1)
            flags = L.DMF_PTR_DEVICE
            if getattr(alpha0, "in_unit_range", None) is True:  # (checked on the host by the thread that uploaded it)
                flags |= L.DMF_INIT_IN_UNIT_RANGE
        else:
            u0 = np.ascontiguousarray(u0, dtype=np.float64).reshape(problem.N, -1)
            alpha0 = np.ascontiguousarray(alpha0, dtype=np.float64)
        self.n_u = int(u0.shape[1])
        self.K = problem.n_c + self.n_u
        if tuple(alpha0.shape) != (self.K, problem.S):
            raise ValueError(f"alpha shape {tuple(alpha0.shape)} != {(self.K, problem.S)}")
        h = C.c_void_p()
        L.check(self._lib.dmf_solver_create(problem.ctx._h, problem._h, _ptr(u0), _ptr(alpha0), self.n_u,
                                            int(mode), flags, C.byref(h)), "dmf_solver_create")
        self._h = h

    def set_purity(self, purity):
        """Switch the alpha phase to the purity-constrained Frank-Wolfe update (deconvolution.py:280-302):
        per-sample mass of the known block, S values in [0, 1]."""
        purity = np.ascontiguousarray(purity, dtype=np.float64).ravel()
        if purity.shape != (self.problem.S,):
            raise ValueError(f"purity needs one value per sample ({self.problem.S}), got {purity.shape}")
        L.check(self._lib.dmf_solver_set_purity(self._h, _ptr(purity), 0), "dmf_solver_set_purity")

    def step(self, n_outer: int, n_iter2: int, tol: float):
        """Run up to n_outer outer iterations; returns (total iterations so far, converged)."""
        it, conv = C.c_int64(), C.c_int()
        L.check(self._lib.dmf_solver_step(self._h, int(n_outer), int(n_iter2), float(tol), C.byref(it),
                                          C.byref(conv)), "dmf_solver_step")
        return it.value, bool(conv.value)

    def get(self):
        """(u, alpha, cost, iterations) of the current iterate, as fresh host arrays."""
        u = np.empty((self.problem.N, self.n_u), dtype=np.float64)
        alpha = np.empty((self.K, self.problem.S), dtype=np.float64)
        cost, it = C.c_double(), C.c_int64()
        L.check(self._lib.dmf_solver_get(self._h, 0, _ptr(u), _ptr(alpha), C.byref(cost), C.byref(it)),
                "dmf_solver_get")
        return u, alpha, cost.value, it.value

    def direct_cost(self) -> float:
        """cost_f_w of the current iterate by the streaming formula (deconvolution.py:15-17), computed where the
        iterate lives: what demethify.py:169,199 and ic.py:206 recompute after a solve."""
        out = C.c_double()
        L.check(self._lib.dmf_solver_cost(self._h, C.byref(out)), "dmf_solver_cost")
        return out.value

    def cost_begin(self):
        """Enqueue direct_cost() without waiting for it (dmf_solver_cost_begin): set up the next solver, then cost_end()."""
        L.check(self._lib.dmf_solver_cost_begin(self._h), "dmf_solver_cost_begin")

    def cost_end(self) -> float:
        out = C.c_double()
        L.check(self._lib.dmf_solver_cost_end(self._h, C.byref(out)), "dmf_solver_cost_end")
        return out.value

    def describe(self, n_iter2: int = 20) -> str:
        """Which kernels a step with n_iter2 inner iterations launches (dmf_solver_describe)."""
        buf = C.create_string_buffer(512)
        L.check(self._lib.dmf_solver_describe(self._h, int(n_iter2), buf, len(buf)), "dmf_solver_describe")
        return buf.value.decode()

    def stop_info(self):
        """How the stop tests of this solver's step() calls were decided (dmf_solver_stop_info): a dict with
        confirm_stops (streaming-cost confirmation active for the last step() call), n_confirmed, n_unconfirmed and
        last_stream_cost (NaN: none taken)."""
        on, nc, nu, cs = C.c_int(), C.c_int64(), C.c_int64(), C.c_double()
        L.check(self._lib.dmf_solver_stop_info(self._h, C.byref(on), C.byref(nc), C.byref(nu), C.byref(cs)),
                "dmf_solver_stop_info")
        return {"confirm_stops": bool(on.value), "n_confirmed": nc.value, "n_unconfirmed": nu.value,
                "last_stream_cost": cs.value}

    def get_alpha(self):
        """The current proportions (K x S) as a fresh host array; u stays on the device."""
        alpha = np.empty((self.K, self.problem.S), dtype=np.float64)
        L.check(self._lib.dmf_solver_get(self._h, 0, None, _ptr(alpha), None, None), "dmf_solver_get")
        return alpha

    def copy_u_to(self, tensor):
        """Copy the current profile estimate u (N x n_u, C order) into a float64 CUDA torch tensor of N * n_u elements
        on the context's GPU (device to device): the bootstrap keeps its replicate stack in HBM."""
        if not (_is_torch(tensor) and tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == self.problem.N * self.n_u
                and tensor.element_size() == 8 and tensor.device.index == self.problem.ctx.device):
            raise ValueError("copy_u_to needs a contiguous float64 CUDA tensor of N * n_u elements on the context's GPU")
        L.check(self._lib.dmf_solver_get(self._h, L.DMF_PTR_DEVICE, _ptr(tensor), None, None, None), "dmf_solver_get")

    def get_cost(self):
        """(cost, iterations) of the current iterate without copying u / alpha back."""
        cost, it = C.c_double(), C.c_int64()
        L.check(self._lib.dmf_solver_get(self._h, 0, None, None, C.byref(cost), C.byref(it)), "dmf_solver_get")
        return cost.value, it.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dmf_solver_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
