"""ctypes binding of libgpcore.so (C ABI in include/gpcore.h).

There is NO CPU fallback: if the shared library is missing, or no HIP device is
visible, every compute entry point raises ``RuntimeError``.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPYREG_AMD_LIB: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("GPYREG_AMD_LIB") or os.path.join(_HERE, "lib", "libgpcore.so")

K_SE, K_MATERN, K_RQ, K_SE_ISO, K_MATERN_ISO = range(5)
F64, F32 = 0, 1
KLO_ZERO, KLO_ROW, KLO_COL = 0, 1, 2
KHI_FULL, KHI_ROW, KHI_COL = 0, 1, 2

# Array arguments are declared void* and passed as the integer address of the NumPy buffer: `a.ctypes.data_as(POINTER(..))`
# costs ~2 us per argument, the address ~1 us, and an evaluation has a dozen of them -- a visible share of a 90 us call.
_dp = C.c_void_p  # double*
_ip = C.c_void_p  # int*
_vp = C.c_void_p
_dpp = C.POINTER(C.c_double)  # typed, for the dK callback (its argument is wrapped as an array)

# every symbol include/gpcore.h declares: (restype, argtypes)
SIGNATURES = {
    "gpc_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "gpc_destroy": (None, [_vp]),
    "gpc_last_error": (C.c_char_p, [_vp]),
    "gpc_device_info": (C.c_char_p, [_vp]),
    "gpc_set_data": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int]),
    "gpc_cov_count": (C.c_int, [C.c_int, C.c_int]),
    "gpc_max_n": (C.c_int, [C.c_int]),
    "gpc_kernel": (
        C.c_int,
        [_vp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp],
    ),
    "gpc_nll_batch": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int,
         _dp, C.c_int, _dp, _dp, _dp, _ip, _ip],
    ),
    "gpc_nll_batch_cm": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int,
         _dp, _dp, _dp, _ip, _ip],
    ),
    "gpc_posterior_batch": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.POINTER(_vp), _dp, _ip, _ip],
    ),
    "gpc_nll_batch_K": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, _dp, _vp, _vp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp,
         C.c_int, _dp, _dp, _dp, _ip, _ip],
    ),
    "gpc_posterior_batch_K": (
        C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.POINTER(_vp), _dp, _ip, _ip]),
    "gpc_predict_K": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "gpc_post_fetch": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "gpc_post_free": (C.c_int, [_vp]),
    "gpc_predict": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp]),
    "gpc_post_append": (C.c_int, [_vp, _dp, _dp, C.c_double, _ip]),
    "gpc_post_recompute": (C.c_int, [_vp, C.c_int, _ip, _dp, _dp, _dp, C.c_int, _dp, _ip, _ip]),
    "gpc_post_append_K": (C.c_int, [_vp, _dp, _dp, _dp, _dp, C.c_double, _ip]),
    "gpc_post_recompute_K": (C.c_int, [_vp, C.c_int, _ip, _dp, _dp, _dp, C.c_int, _dp, _ip, _ip]),
    "gpc_predict_full": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp]),
    "gpc_quad": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]),
    "gpc_last_timing": (C.c_int, [_vp, _dp, _dp]),
    "gpc_last_lauum_timing": (C.c_int, [_vp, _dp, _dp]),
    "gpc_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "gpc_get_option": (C.c_int, [_vp, C.c_char_p, _ip]),
    "gpc_mfma_peak": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "gpc_debug_gemm": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
         C.c_int, C.c_int, _dp, _dp, _dp],
    ),
    "gpc_debug_leaf": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _ip]),
    "gpc_debug_factor": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpc_debug_workspace_hash": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
}
# declared under GPC_EXPERIMENTS in include/gpcore.h: present in the experiments build only (lib/libgpcore_exp.so, which
# tests/ and tools/ select through GPYREG_AMD_LIB; the product library does not export them)
EXPERIMENT_SIGNATURES = {
    "gpc_debug_dag": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, _ip, C.c_int, C.c_int]),
}
EXPERIMENTS_LIB_PATH = os.path.join(_HERE, "lib", "libgpcore_exp.so")

# int (*gpc_dk_plane_fn)(void* user, int sample, int p, double* plane)
DK_PLANE_FN = C.CFUNCTYPE(C.c_int, _vp, C.c_int, C.c_int, _dpp)

_lib = None
_lock = threading.RLock()


def load():
    """Load libgpcore.so and attach prototypes.  Raises RuntimeError when missing."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: the HIP extension has not been built "
                    "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                    "gpyreg_amd has no CPU fallback."
                )
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so does not export it
                fn.restype = res
                fn.argtypes = args
            for name, (res, args) in EXPERIMENT_SIGNATURES.items():
                fn = getattr(lib, name, None)
                if fn is not None:
                    fn.restype = res
                    fn.argtypes = args
            _lib = lib
    return _lib


def is_experiments_build() -> bool:
    """True when the loaded library is the experiments build (``GPYREG_AMD_LIB=.../libgpcore_exp.so``)."""
    return hasattr(load(), "gpc_debug_dag")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    """The bare address of an array for a ``c_void_p`` argument.  KEEP-ALIVE INVARIANT: unlike ``ctypes.data_as`` the
    integer holds no reference to the array, and the call it is passed to releases the GIL -- so ``a`` must be bound to
    a name that outlives the call (every call site converts into a local first: ``x = _f64(x)`` ... ``_ptr(x)``).
    Never ``_ptr(_f64(x))`` or ``_ptr(x.ravel())`` inline: the temporary would be freed before the library reads it."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags.c_contiguous, "pass a named, C-contiguous ndarray (see docstring)"
    return a.ctypes.data


def _serial(method):
    """A gpc_ctx is one stream with one workspace: calls on it must not overlap (include/gpcore.h), and ctypes releases
    the GIL while a call runs.  Every entry of a Context and of its PostHandles takes the context's re-entrant lock."""
    import functools

    @functools.wraps(method)
    def locked(self, *a, **k):
        with (self.lock if isinstance(self, Context) else self.ctx.lock):
            return method(self, *a, **k)

    return locked


class Context:
    """One gpc_ctx (device stream + workspace + resident X, y)."""

    def __init__(self, device: int = 0):
        self.lock = threading.RLock()
        self._lib = load()
        h = _vp()
        rc = self._lib.gpc_create(int(device), C.byref(h))
        if rc != 0:
            msg = self._lib.gpc_last_error(None).decode()
            raise RuntimeError(f"gpc_create(device={device}) failed: {msg} (no CPU fallback)")
        self._h = h
        self.device = int(device)
        self.data_token = None
        self.N = self.D = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpc_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed (rc={rc}): {self._lib.gpc_last_error(self._h).decode()}")

    def device_info(self) -> str:
        return self._lib.gpc_device_info(self._h).decode()

    # ---- data -----------------------------------------------------------------
    @_serial
    def set_data(self, X, y, token=None):
        X = _f64(X)
        y = _f64(y).ravel()
        N, D = X.shape
        if y.size != N:  # (the library reads N values of y: a shorter array would be read past its end)
            raise ValueError(f"X has {N} points and y has {y.size}: every input needs its observation")
        self._check(self._lib.gpc_set_data(self._h, _ptr(X), _ptr(y), N, D), "gpc_set_data")
        self.N, self.D = N, D
        self.data_token = token

    # ---- covariance.compute ---------------------------------------------------
    @_serial
    def kernel(self, kid, degree, hyp, X, X_star=None, diag=False, grad=False):
        X = _f64(X)
        hyp = _f64(hyp)
        N, D = X.shape
        cov_N = self._lib.gpc_cov_count(kid, D)
        Xs = None if X_star is None else _f64(X_star)
        M = 0 if Xs is None else Xs.shape[0]
        if diag:
            K = np.empty((N, 1))
        else:
            K = np.empty((N, M if Xs is not None else N))
        dK = np.empty((N, N, cov_N)) if grad else None
        rc = self._lib.gpc_kernel(self._h, kid, degree, _ptr(hyp), _ptr(X), N, D, _ptr(Xs), M,
                                  1 if diag else 0, _ptr(K), _ptr(dK))
        self._check(rc, "gpc_kernel")
        return (K, dK) if grad else K

    # ---- core -----------------------------------------------------------------
    @_serial
    def nll_batch(self, kid, degree, dtype, hyp_cov, m, sn2, sn2_is_vector, want_grad=False,
                  dm=None, dsn2=None):
        """hyp_cov (S,cov_N); m (S,N); sn2 (S,N) if sn2_is_vector else (S,1);
        dm (S,N,mean_N); dsn2 (S, N if sn2_is_vector else 1, noise_N)."""
        hyp_cov = _f64(hyp_cov)
        m = _f64(m)
        sn2 = _f64(sn2)
        S, cov_N = hyp_cov.shape
        vec = 1 if sn2_is_vector else 0
        if sn2.shape != (S, self.N if vec else 1) or m.shape != (S, self.N):
            raise ValueError("m must be (S,N); sn2 must be (S,N) when per-point, else (S,1)")
        mean_N = 0 if dm is None else dm.shape[2]
        noise_N = 0 if dsn2 is None else dsn2.shape[2]
        dm_c = None if dm is None or mean_N == 0 else _f64(dm)
        dsn2_c = None if dsn2 is None or noise_N == 0 else _f64(dsn2)
        hyp_N = cov_N + noise_N + mean_N
        nlz = np.empty(S)
        dnlz = np.empty((S, hyp_N)) if want_grad else None
        mult = np.empty(S)
        lchol = np.empty(S, dtype=np.int32)
        info = np.empty(S, dtype=np.int32)
        rc = self._lib.gpc_nll_batch(
            self._h, kid, degree, dtype, S, _ptr(hyp_cov), _ptr(m), _ptr(sn2), vec,
            1 if want_grad else 0, _ptr(dm_c), mean_N, _ptr(dsn2_c), noise_N, _ptr(nlz), _ptr(dnlz),
            _ptr(mult), lchol.ctypes.data, info.ctypes.data)
        self._check(rc, "gpc_nll_batch")
        return nlz, dnlz, mult, lchol.astype(bool), info

    @_serial
    def nll_batch_cm(self, kid, degree, dtype, hyp_cov, m0, sn2, sn2_is_vector, want_grad=False, dsn2=None):
        """gpc_nll_batch_cm: the stock zero / constant mean as ONE value per sample -- m0 (S,) or None for the zero
        mean; dsn2 (S, N if sn2_is_vector else 1, noise_N).  Same results as ``nll_batch`` with m = m0 and dm = 1."""
        hyp_cov = _f64(hyp_cov)
        sn2 = _f64(sn2)
        S, cov_N = hyp_cov.shape
        vec = 1 if sn2_is_vector else 0
        mean_N = 0 if m0 is None else 1
        m0_c = None if m0 is None else _f64(m0).ravel()
        if sn2.shape != (S, self.N if vec else 1) or (m0_c is not None and m0_c.shape != (S,)):
            raise ValueError("m0 must be (S,); sn2 must be (S,N) when per-point, else (S,1)")
        noise_N = 0 if dsn2 is None else dsn2.shape[2]
        dsn2_c = None if dsn2 is None or noise_N == 0 else _f64(dsn2)
        hyp_N = cov_N + noise_N + mean_N
        nlz = np.empty(S)
        dnlz = np.empty((S, hyp_N)) if want_grad else None
        mult = np.empty(S)
        lchol = np.empty(S, dtype=np.int32)
        info = np.empty(S, dtype=np.int32)
        rc = self._lib.gpc_nll_batch_cm(
            self._h, kid, degree, dtype, S, _ptr(hyp_cov), _ptr(m0_c), mean_N, _ptr(sn2), vec,
            1 if want_grad else 0, _ptr(dsn2_c), noise_N, _ptr(nlz), _ptr(dnlz), _ptr(mult), lchol.ctypes.data,
            info.ctypes.data)
        self._check(rc, "gpc_nll_batch_cm")
        return nlz, dnlz, mult, lchol.astype(bool), info

    @_serial
    def nll_batch_K(self, dtype, K, dK_plane, cov_N, m, sn2, sn2_is_vector, want_grad=False, dm=None,
                    dsn2=None):
        """gpc_nll_batch_K: K (S,N,N) from the caller's covariance object; ``dK_plane(s, p)`` returns
        the (N,N) array dK[:, :, p] of sample s (called once per sample and hyperparameter)."""
        K = _f64(K)
        m = _f64(m)
        sn2 = _f64(sn2)
        S, N = K.shape[0], K.shape[1]
        vec = 1 if sn2_is_vector else 0
        if K.shape != (S, self.N, self.N) or sn2.shape != (S, self.N if vec else 1) or m.shape != (S, self.N):
            raise ValueError("K must be (S,N,N); m (S,N); sn2 (S,N) when per-point, else (S,1)")
        mean_N = 0 if dm is None else dm.shape[2]
        noise_N = 0 if dsn2 is None else dsn2.shape[2]
        dm_c = None if dm is None or mean_N == 0 else _f64(dm)
        dsn2_c = None if dsn2 is None or noise_N == 0 else _f64(dsn2)
        hyp_N = cov_N + noise_N + mean_N
        nlz = np.empty(S)
        dnlz = np.empty((S, hyp_N)) if want_grad else None
        mult = np.empty(S)
        lchol = np.empty(S, dtype=np.int32)
        info = np.empty(S, dtype=np.int32)
        err = []

        def plane_cb(_user, s, p, out):
            try:
                np.ctypeslib.as_array(out, shape=(N, N))[...] = dK_plane(s, p)
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the return code, re-raised below
                err.append(e)
                return 1

        cb = DK_PLANE_FN(plane_cb)
        rc = self._lib.gpc_nll_batch_K(
            self._h, dtype, S, cov_N, _ptr(K), C.cast(cb, _vp) if want_grad else None, None, _ptr(m),
            _ptr(sn2), vec, 1 if want_grad else 0, _ptr(dm_c), mean_N, _ptr(dsn2_c), noise_N, _ptr(nlz),
            _ptr(dnlz), _ptr(mult), lchol.ctypes.data, info.ctypes.data)
        if err:
            raise err[0]
        self._check(rc, "gpc_nll_batch_K")
        return nlz, dnlz, mult, lchol.astype(bool), info

    @_serial
    def posterior_batch_K(self, dtype, K, m, sn2, sn2_is_vector):
        K, m, sn2 = _f64(K), _f64(m), _f64(sn2)
        S = K.shape[0]
        vec = 1 if sn2_is_vector else 0
        if K.shape != (S, self.N, self.N) or sn2.shape != (S, self.N if vec else 1) or m.shape != (S, self.N):
            raise ValueError("K must be (S,N,N); m (S,N); sn2 (S,N) when per-point, else (S,1)")
        mult = np.empty(S)
        lchol = np.empty(S, dtype=np.int32)
        info = np.empty(S, dtype=np.int32)
        h = _vp()
        rc = self._lib.gpc_posterior_batch_K(self._h, dtype, S, _ptr(K), _ptr(m), _ptr(sn2), vec, C.byref(h),
                                             _ptr(mult), lchol.ctypes.data, info.ctypes.data)
        self._check(rc, "gpc_posterior_batch_K")
        return PostHandle(self, h, S, self.N), mult, lchol.astype(bool), info

    @_serial
    def posterior_batch(self, kid, degree, dtype, hyp_cov, m, sn2, sn2_is_vector):
        hyp_cov = _f64(hyp_cov)
        m = _f64(m)
        sn2 = _f64(sn2)
        S = hyp_cov.shape[0]
        vec = 1 if sn2_is_vector else 0
        if sn2.shape != (S, self.N if vec else 1) or m.shape != (S, self.N):
            raise ValueError("m must be (S,N); sn2 must be (S,N) when per-point, else (S,1)")
        mult = np.empty(S)
        lchol = np.empty(S, dtype=np.int32)
        info = np.empty(S, dtype=np.int32)
        h = _vp()
        rc = self._lib.gpc_posterior_batch(
            self._h, kid, degree, dtype, S, _ptr(hyp_cov), _ptr(m), _ptr(sn2), vec, C.byref(h),
            _ptr(mult), lchol.ctypes.data, info.ctypes.data)
        self._check(rc, "gpc_posterior_batch")
        return PostHandle(self, h, S, self.N), mult, lchol.astype(bool), info

    @_serial
    def last_timing(self):
        a, b = C.c_double(), C.c_double()
        self._lib.gpc_last_timing(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    @_serial
    def set_option(self, name: str, value: int):
        self._check(self._lib.gpc_set_option(self._h, name.encode(), int(value)), "gpc_set_option")

    @_serial
    def get_option(self, name: str) -> int:
        v = C.c_int()
        self._check(self._lib.gpc_get_option(self._h, name.encode(), C.byref(v)), "gpc_get_option")
        return v.value

    @_serial
    def last_lauum_timing(self):
        a, b = C.c_double(), C.c_double()
        self._lib.gpc_last_lauum_timing(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    @_serial
    def mfma_peak(self, dtype=F64):
        """(TFLOP/s, shader cycles per MFMA per SIMD, clock in GHz) of a bare MFMA loop."""
        t, cyc, ghz = C.c_double(), C.c_double(), C.c_double()
        rc = self._lib.gpc_mfma_peak(self._h, dtype, C.byref(t), C.byref(cyc), C.byref(ghz))
        self._check(rc, "gpc_mfma_peak")
        return t.value, cyc.value, ghz.value

    # ---- test hooks -------------------------------------------------------------
    @_serial
    def debug_gemm(self, A, B, Cm, M, N, K, a_kmajor, b_kmajor, alpha=1.0, beta=0, klo=0, khi=0,
                   lower_only=False, dtype=F64, force_bt=0):
        A, B = _f64(A), _f64(B)
        Cm = _f64(Cm).copy()
        flags = int(bool(lower_only)) | {0: 0, 64: 0x100, 128: 0x200, 12864: 0x400}[force_bt]  # (12864: experiments build)
        rc = self._lib.gpc_debug_gemm(self._h, dtype, M, N, K, int(a_kmajor), int(b_kmajor),
                                      float(alpha), int(beta), klo, khi, flags, _ptr(A),
                                      _ptr(B), _ptr(Cm))
        self._check(rc, "gpc_debug_gemm")
        return Cm

    @_serial
    def debug_factor(self, A, want_inv=True, dtype=F64):
        A = _f64(A)
        n = A.shape[0]
        L, W = np.empty((n, n)), np.empty((n, n))
        Ainv = np.empty((n, n)) if want_inv else None
        logdet = C.c_double()
        info = C.c_int()
        rc = self._lib.gpc_debug_factor(self._h, dtype, n, _ptr(A), _ptr(L), _ptr(W), _ptr(Ainv),
                                        C.byref(logdet), C.byref(info))
        self._check(rc, "gpc_debug_factor")
        return L, W, Ainv, logdet.value, info.value


class PostHandle:
    """Device-resident posteriors of one hyperparameter batch (gpc_post)."""

    def __init__(self, ctx: Context, h, S: int, N: int):
        self.ctx, self._h, self.S, self.N = ctx, h, S, N

    @_serial
    def fetch(self, s, alpha=True, sW=True, L=True):
        N = self.N
        a = np.empty(N) if alpha else None
        w = np.empty(N) if sW else None
        Lm = np.empty((N, N)) if L else None
        rc = self.ctx._lib.gpc_post_fetch(self._h, int(s), _ptr(a), _ptr(w), _ptr(Lm))
        self.ctx._check(rc, "gpc_post_fetch")
        return a, w, Lm

    @_serial
    def predict(self, x_star):
        xs = _f64(x_star)
        M = xs.shape[0]
        fmu = np.empty((M, self.S))
        fs2 = np.empty((M, self.S))
        rc = self.ctx._lib.gpc_predict(self._h, _ptr(xs), M, _ptr(fmu), _ptr(fs2))
        self.ctx._check(rc, "gpc_predict")
        return fmu, fs2

    @_serial
    def predict_K(self, Ks, Kss=None, want_var=True):
        """gpc_predict_K: Ks (S,N,M) caller-provided cross covariances; returns fmu (M,S), the
        variance term fq (M,S; add kss) and, with Kss (S,M,M), the full covariances (S,M,M)."""
        Ks = _f64(Ks)
        M = Ks.shape[2]
        fmu = np.empty((M, self.S))
        fq = np.empty((M, self.S)) if want_var else None
        Kss_c = None if Kss is None else _f64(Kss)
        cov = None if Kss is None else np.empty((self.S, M, M))
        rc = self.ctx._lib.gpc_predict_K(self._h, M, _ptr(Ks), _ptr(Kss_c), _ptr(fmu), _ptr(fq), _ptr(cov))
        self.ctx._check(rc, "gpc_predict_K")
        return fmu, fq, cov

    @_serial
    def append(self, m_star, sn2_star, y_new):
        """Rank-one append of the point already added to the context's data.  Returns the
        per-sample outcome (bool array): False entries must be recomputed (``recompute``)."""
        m_star, sn2_star = _f64(m_star).ravel(), _f64(sn2_star).ravel()
        ok = np.zeros(self.S, dtype=np.int32)
        rc = self.ctx._lib.gpc_post_append(self._h, _ptr(m_star), _ptr(sn2_star), float(y_new),
                                           ok.ctypes.data)
        self.ctx._check(rc, "gpc_post_append")
        self.N += 1
        return ok.astype(bool)

    @_serial
    def append_K(self, Ks, kss, m_star, sn2_star, y_new):
        """``append`` for posteriors built from a caller's covariance object: Ks (S, n) = k_s(X_old, x_new),
        kss (S,) = k_s(x_new, x_new)."""
        Ks, kss = _f64(Ks), _f64(kss).ravel()
        m_star, sn2_star = _f64(m_star).ravel(), _f64(sn2_star).ravel()
        if Ks.shape != (self.S, self.N) or kss.shape != (self.S,):
            raise ValueError("Ks must be (S, n) and kss (S,)")
        ok = np.zeros(self.S, dtype=np.int32)
        rc = self.ctx._lib.gpc_post_append_K(self._h, _ptr(Ks), _ptr(kss), _ptr(m_star), _ptr(sn2_star),
                                             float(y_new), ok.ctypes.data)
        self.ctx._check(rc, "gpc_post_append_K")
        self.N += 1
        return ok.astype(bool)

    @_serial
    def recompute(self, idx, hyp_cov, m, sn2, sn2_is_vector, K=None):
        """Full recompute of the listed samples in place (the reference's ``full_updates``).  ``K`` (cnt, N, N):
        the caller's covariance matrices on the extended data (posteriors built from a covariance object)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        m, sn2 = _f64(m), _f64(sn2)
        cnt = idx.size
        mult = np.empty(cnt)
        lchol = np.empty(cnt, dtype=np.int32)
        info = np.empty(cnt, dtype=np.int32)
        if K is not None:
            K = _f64(K)
            rc = self.ctx._lib.gpc_post_recompute_K(
                self._h, cnt, idx.ctypes.data, _ptr(K), _ptr(m), _ptr(sn2),
                1 if sn2_is_vector else 0, _ptr(mult), lchol.ctypes.data, info.ctypes.data)
            self.ctx._check(rc, "gpc_post_recompute_K")
            return mult, lchol.astype(bool), info
        hyp_cov = _f64(hyp_cov)
        rc = self.ctx._lib.gpc_post_recompute(
            self._h, cnt, idx.ctypes.data, _ptr(hyp_cov), _ptr(m), _ptr(sn2),
            1 if sn2_is_vector else 0, _ptr(mult), lchol.ctypes.data, info.ctypes.data)
        self.ctx._check(rc, "gpc_post_recompute")
        return mult, lchol.astype(bool), info

    @_serial
    def predict_full(self, x_star):
        xs = _f64(x_star)
        M = xs.shape[0]
        fmu = np.empty((M, self.S))
        cov = np.empty((self.S, M, M))
        rc = self.ctx._lib.gpc_predict_full(self._h, _ptr(xs), M, _ptr(fmu), _ptr(cov))
        self.ctx._check(rc, "gpc_predict_full")
        return fmu, cov

    @_serial
    def quad(self, mu, sigma, compute_var):
        mu, sigma = _f64(mu), _f64(sigma)
        M = mu.shape[0]
        za = np.empty((M, self.S))
        zkz = np.empty((M, self.S)) if compute_var else None
        rc = self.ctx._lib.gpc_quad(self._h, _ptr(mu), _ptr(sigma), M, 1 if compute_var else 0, _ptr(za), _ptr(zkz))
        self.ctx._check(rc, "gpc_quad")
        return za, zkz

    @_serial
    def free(self):
        if self._h:
            self.ctx._lib.gpc_post_free(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            if self.ctx._h:
                self.free()
        except Exception:
            pass


_contexts = {}


def default_device() -> int:
    return int(os.environ.get("GPYREG_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def context(device: int | None = None) -> Context:
    """Process-wide context of a device (created on first use)."""
    dev = default_device() if device is None else int(device)
    with _lock:
        ctx = _contexts.get(dev)
        if ctx is None or ctx._h is None:
            ctx = Context(dev)
            _contexts[dev] = ctx
    return ctx
