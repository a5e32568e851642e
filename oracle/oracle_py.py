"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

``Oracle``   -> oracle/_build/libpcroracle.so  (the plain-C restatement, pcr_oracle.c)
``RefShim``  -> oracle/_ref/libpcrref.so       (the UNMODIFIED reference objects behind
                                                 ref_shim.cpp; exists only where
                                                 /root/reference was available at build time)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  Nothing under primalcr_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libpcroracle.so")
REF_SO = os.path.join(HERE, "_ref", "libpcrref.so")
REF_TRAIN = os.path.join(HERE, "_ref", "omp-pmf-train")
REF_PREDICT = os.path.join(HERE, "_ref", "omp-pmf-predict")

_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(target="oracle"):
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class CSR:
    """User-major CSR in the reference's SparseMat layout (util.h:390-413)."""

    def __init__(self, d1, d2, idx, item, val):
        self.d1, self.d2 = int(d1), int(d2)
        self.idx, self.item, self.val = i64(idx), i64(item), f64(val)

    @property
    def nnz(self):
        return int(self.idx[self.d1])


class IterRec(C.Structure):
    _fields_ = [("obj", C.c_double), ("train_err", C.c_double), ("train_ndcg", C.c_double),
                ("test_err", C.c_double), ("test_ndcg", C.c_double), ("seconds", C.c_double),
                ("cg_v", C.c_long), ("ls_v", C.c_long), ("cg_u", C.c_long), ("ls_u", C.c_long)]


class Oracle:
    def __init__(self, path=None):
        path = path or ORACLE_SO                 # (read at call time: tests/conftest.py points it at the sanitizer build)
        if not os.path.exists(path):
            build("oracle")
        self.lib = L = C.CDLL(path)
        L.orc_initial.argtypes = [_dp, C.c_long, C.c_long]
        L.orc_build_csr.argtypes = [C.c_long, C.c_long, _ip, _ip, _dp, _lp, _lp, _dp]
        L.orc_build_csr_test.argtypes = [C.c_long, C.c_long, _ip, _ip, _dp, _lp, _lp, _dp]
        L.orc_build_csr_test.restype = C.c_long
        L.orc_comp_m.argtypes = [_dp, _dp, C.c_long, _lp, _lp, C.c_int, _dp]
        for name in ("orc_objective_new", "orc_objective"):
            f = getattr(L, name)
            f.argtypes = [_dp, _dp, _dp, C.c_long, C.c_long, _lp, _dp, C.c_int, C.c_double]
            f.restype = C.c_double
        for name in ("orc_obtain_g_new", "orc_obtain_g"):
            getattr(L, name).argtypes = [_dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, _dp, C.c_int, C.c_double, _dp]
        for name in ("orc_compute_Ha_new", "orc_compute_Ha"):
            getattr(L, name).argtypes = [_dp, _dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, C.c_int, C.c_double, _dp]
        L.orc_solve_delta_new.argtypes = [_dp, _dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, C.c_int, C.c_double, _dp]
        L.orc_solve_delta_new.restype = C.c_int
        L.orc_update_V_new.argtypes = [C.c_long, C.c_long, _lp, _lp, _dp, C.c_double, C.c_double, C.c_int,
                                       _dp, _dp, C.POINTER(C.c_double), _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_update_V_new.restype = C.c_int
        for name in ("orc_update_u_new", "orc_update_u"):
            f = getattr(L, name)
            f.argtypes = [C.c_long, _dp, _lp, _lp, _dp, _dp, C.c_int, C.c_double, C.c_double, _dp, _dp,
                          C.POINTER(C.c_double), C.POINTER(C.c_int)]
            f.restype = C.c_int
        L.orc_update_U_new.argtypes = [C.c_long, C.c_long, _lp, _lp, _dp, _dp, C.c_double, C.c_double, C.c_int,
                                       _dp, _dp, _dp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.orc_eval.argtypes = [_dp, _dp, C.c_long, _lp, _lp, _dp, C.c_int, C.c_int,
                               C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_train.argtypes = [C.c_int, C.c_long, C.c_long, _lp, _lp, _dp, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_long, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double,
                                _dp, _dp, C.POINTER(IterRec)]
        L.orc_count_pairs.argtypes = [C.c_long, _lp, _dp, C.c_int]
        L.orc_count_pairs.restype = C.c_long

    # ---- host-side helpers -------------------------------------------------
    def initial(self, n, k):
        X = np.empty((n, k), np.float64)
        self.lib.orc_initial(X, n, k)
        return X

    def set_cg(self, max_iter=10, tol=0.01):
        """CG cap / tolerance of every solve (the reference's constants by default); process-global."""
        self.lib.orc_set_cg.argtypes = [C.c_int, C.c_double]
        self.lib.orc_set_cg(max_iter, tol)

    def build_csr(self, d1, d2, user, item, val):
        nnz = len(user)
        idx = np.empty(d1 + 1, np.int64); it = np.empty(nnz, np.int64); v = np.empty(nnz, np.float64)
        self.lib.orc_build_csr(d1, nnz, np.ascontiguousarray(user, np.int32),
                               np.ascontiguousarray(item, np.int32), f64(val), idx, it, v)
        return CSR(d1, d2, idx, it, v)

    def build_csr_test(self, d1, d2, user, item, val):
        nnz = len(user)
        idx = np.empty(d1 + 1, np.int64); it = np.zeros(max(nnz, 1), np.int64); v = np.zeros(max(nnz, 1), np.float64)
        n = self.lib.orc_build_csr_test(d1, nnz, np.ascontiguousarray(user, np.int32),
                                        np.ascontiguousarray(item, np.int32), f64(val), idx, it, v)
        return CSR(d1, d2, idx, it[:n], v[:n])

    # ---- PrimalCR++ ---------------------------------------------------------
    def comp_m(self, U, V, X):
        m = np.empty(max(X.nnz, 1), np.float64)
        self.lib.orc_comp_m(f64(U), f64(V), X.d1, X.idx, X.item, U.shape[1], m)
        return m[:X.nnz]

    def objective_new(self, m, U, V, X, lam, solver=2):
        f = self.lib.orc_objective_new if solver == 2 else self.lib.orc_objective
        return f(f64(m), f64(U), f64(V), X.d1, X.d2, X.idx, X.val, U.shape[1], lam)

    def obtain_g_new(self, U, V, X, m, lam, solver=2):
        g = np.empty_like(f64(V))
        f = self.lib.orc_obtain_g_new if solver == 2 else self.lib.orc_obtain_g
        f(f64(U), f64(V), X.d1, X.d2, X.idx, X.item, X.val, f64(m), U.shape[1], lam, g)
        return g

    def compute_Ha_new(self, a, m, U, X, lam, solver=2):
        Ha = np.empty_like(f64(a))
        f = self.lib.orc_compute_Ha_new if solver == 2 else self.lib.orc_compute_Ha
        f(f64(a), f64(m), f64(U), X.d1, X.d2, X.idx, X.item, X.val, U.shape[1], lam, Ha)
        return Ha

    def solve_delta_new(self, g, m, U, X, lam):
        delta = np.empty_like(f64(g))
        its = self.lib.orc_solve_delta_new(f64(g), f64(m), f64(U), X.d1, X.d2, X.idx, X.item, X.val,
                                           U.shape[1], lam, delta)
        return delta, its

    def update_V_new(self, X, lam, stepsize, U, V):
        V = f64(V).copy()
        m = np.empty(max(X.nnz, 1), np.float64)
        obj = C.c_double(); acc = C.c_int(); cg = C.c_int()
        ls = self.lib.orc_update_V_new(X.d1, X.d2, X.idx, X.item, X.val, lam, stepsize, U.shape[1],
                                       f64(U), V, C.byref(obj), m, C.byref(acc), C.byref(cg))
        return V, m[:X.nnz], obj.value, dict(ls=ls, accepted=acc.value, cg=cg.value)

    def update_u_new(self, i, V, X, m, lam, stepsize, ui, solver=2):
        out = np.empty_like(f64(ui))
        obj = C.c_double(); nls = C.c_int()
        f = self.lib.orc_update_u_new if solver == 2 else self.lib.orc_update_u
        cg = f(i, f64(V), X.idx, X.item, X.val, f64(m), V.shape[1], lam, stepsize, f64(ui), out,
               C.byref(obj), C.byref(nls))
        return out, obj.value, dict(cg=cg, ls=nls.value)

    def update_U_new(self, X, m, lam, stepsize, V, U):
        Un = np.empty_like(f64(U))
        obj = C.c_double(); tcg = C.c_long(); tls = C.c_long()
        self.lib.orc_update_U_new(X.d1, X.d2, X.idx, X.item, X.val, f64(m), lam, stepsize, U.shape[1],
                                  f64(V), f64(U), Un, C.byref(obj), C.byref(tcg), C.byref(tls))
        return Un, obj.value, dict(cg=tcg.value, ls=tls.value)

    def eval(self, U, V, X, ndcg_k=10):
        e = C.c_double(); n = C.c_double()
        self.lib.orc_eval(f64(U), f64(V), X.d1, X.idx, X.item, X.val, U.shape[1], ndcg_k, C.byref(e), C.byref(n))
        return e.value, n.value

    def train(self, X, U, V, lam, maxiter, XT=None, solver=2, do_predict=1, ndcg_k=10, stepsize=1.0):
        U = f64(U).copy(); V = f64(V).copy()
        hist = (IterRec * (maxiter + 1))()
        if XT is not None and XT.nnz > 0:
            targs = (XT.idx.ctypes.data, XT.item.ctypes.data, XT.val.ctypes.data, XT.nnz)
        else:
            targs = (None, None, None, 0)
        self.lib.orc_train(solver, X.d1, X.d2, X.idx, X.item, X.val, *targs, U.shape[1], lam, maxiter,
                           do_predict, ndcg_k, stepsize, U, V, hist)
        recs = [{k: getattr(h, k) for k, _ in IterRec._fields_} for h in hist]
        return U, V, recs

    def count_pairs(self, X, raw=False):
        return self.lib.orc_count_pairs(X.d1, X.idx, X.val, int(raw))


class RefShim:
    """The compiled, unmodified reference (only available where oracle/_ref was built)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, path=REF_SO, threads=1):
        self.lib = L = C.CDLL(path)
        L.ref_set_threads(threads)
        L.ref_initial.argtypes = [_dp, C.c_long, C.c_long]
        L.ref_load_dir.argtypes = [C.c_char_p] + [C.POINTER(C.c_long)] * 4 + [C.c_void_p] * 6
        L.ref_comp_m_new.argtypes = [_dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, C.c_int, _dp]
        for name in ("ref_objective_new", "ref_objective"):
            f = getattr(L, name)
            f.argtypes = [_dp, _dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, C.c_int, C.c_double]
            f.restype = C.c_double
        for name in ("ref_obtain_g_new", "ref_obtain_g"):
            getattr(L, name).argtypes = [_dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, _dp, C.c_int, C.c_double, _dp]
        for name in ("ref_compute_Ha_new", "ref_compute_Ha", "ref_solve_delta_new"):
            getattr(L, name).argtypes = [_dp, _dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, C.c_int, C.c_double, _dp]
        L.ref_update_V_new.argtypes = [C.c_long, C.c_long, _lp, _lp, _dp, C.c_double, C.c_double, C.c_int,
                                       _dp, _dp, C.POINTER(C.c_double), _dp]
        for name in ("ref_update_u_new", "ref_update_u"):
            getattr(L, name).argtypes = [C.c_long, _dp, C.c_long, C.c_long, _lp, _lp, _dp, _dp, C.c_int,
                                         C.c_double, C.c_double, _dp, _dp, C.POINTER(C.c_double)]
        L.ref_update_U_new.argtypes = [C.c_long, C.c_long, _lp, _lp, _dp, _dp, C.c_double, C.c_double, C.c_int,
                                       _dp, _dp, _dp, C.POINTER(C.c_double)]
        L.ref_eval.argtypes = [_dp, _dp, C.c_long, C.c_long, _lp, _lp, _dp, C.c_int, C.c_int,
                               C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def initial(self, n, k):
        X = np.empty((n, k), np.float64)
        self.lib.ref_initial(X, n, k)
        return X

    def load_dir(self, path):
        d1 = C.c_long(); d2 = C.c_long(); nnz = C.c_long(); tnnz = C.c_long()
        self.lib.ref_load_dir(path.encode(), d1, d2, nnz, tnnz, None, None, None, None, None, None)
        idx = np.empty(d1.value + 1, np.int64); item = np.empty(nnz.value, np.int64); val = np.empty(nnz.value, np.float64)
        tidx = np.empty(d1.value + 1, np.int64); titem = np.zeros(max(tnnz.value, 1), np.int64)
        tval = np.zeros(max(tnnz.value, 1), np.float64)
        self.lib.ref_load_dir(path.encode(), d1, d2, nnz, tnnz, idx.ctypes.data, item.ctypes.data, val.ctypes.data,
                              tidx.ctypes.data, titem.ctypes.data, tval.ctypes.data)
        n = int(tidx[d1.value])
        return CSR(d1.value, d2.value, idx, item, val), CSR(d1.value, d2.value, tidx, titem[:n], tval[:n])

    def comp_m(self, U, V, X):
        m = np.empty(max(X.nnz, 1), np.float64)
        self.lib.ref_comp_m_new(f64(U), f64(V), X.d1, X.d2, X.idx, X.item, X.val, U.shape[1], m)
        return m[:X.nnz]

    def objective_new(self, m, U, V, X, lam, solver=2):
        f = self.lib.ref_objective_new if solver == 2 else self.lib.ref_objective
        return f(f64(m), f64(U), f64(V), X.d1, X.d2, X.idx, X.item, X.val, U.shape[1], lam)

    def obtain_g_new(self, U, V, X, m, lam, solver=2):
        g = np.empty_like(f64(V))
        f = self.lib.ref_obtain_g_new if solver == 2 else self.lib.ref_obtain_g
        f(f64(U), f64(V), X.d1, X.d2, X.idx, X.item, X.val, f64(m), U.shape[1], lam, g)
        return g

    def compute_Ha_new(self, a, m, U, X, lam, solver=2):
        Ha = np.empty_like(f64(a))
        f = self.lib.ref_compute_Ha_new if solver == 2 else self.lib.ref_compute_Ha
        f(f64(a), f64(m), f64(U), X.d1, X.d2, X.idx, X.item, X.val, U.shape[1], lam, Ha)
        return Ha

    def solve_delta_new(self, g, m, U, X, lam):
        delta = np.empty_like(f64(g))
        self.lib.ref_solve_delta_new(f64(g), f64(m), f64(U), X.d1, X.d2, X.idx, X.item, X.val, U.shape[1], lam, delta)
        return delta

    def update_V_new(self, X, lam, stepsize, U, V):
        V = f64(V).copy()
        m = np.empty(max(X.nnz, 1), np.float64)
        obj = C.c_double()
        self.lib.ref_update_V_new(X.d1, X.d2, X.idx, X.item, X.val, lam, stepsize, U.shape[1], f64(U), V,
                                  C.byref(obj), m)
        return V, m[:X.nnz], obj.value

    def update_u_new(self, i, V, X, m, lam, stepsize, ui, solver=2):
        out = np.empty_like(f64(ui))
        obj = C.c_double()
        f = self.lib.ref_update_u_new if solver == 2 else self.lib.ref_update_u
        f(i, f64(V), X.d1, X.d2, X.idx, X.item, X.val, f64(m), V.shape[1], lam, stepsize, f64(ui), out, C.byref(obj))
        return out, obj.value

    def update_U_new(self, X, m, lam, stepsize, V, U):
        Un = np.empty_like(f64(U))
        obj = C.c_double()
        self.lib.ref_update_U_new(X.d1, X.d2, X.idx, X.item, X.val, f64(m), lam, stepsize, U.shape[1],
                                  f64(V), f64(U), Un, C.byref(obj))
        return Un, obj.value

    def eval(self, U, V, X, ndcg_k=10):
        e = C.c_double(); n = C.c_double()
        self.lib.ref_eval(f64(U), f64(V), X.d1, X.d2, X.idx, X.item, X.val, U.shape[1], ndcg_k, C.byref(e), C.byref(n))
        return e.value, n.value
