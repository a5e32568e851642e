"""ctypes mirror of include/primalcr.h (the reference's solver interface, pmf.h:9-56)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PCR_SOLVER_PCR, PCR_SOLVER_PCRPP = 1, 2
PCR_F32, PCR_F64 = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))


_LIB_OVERRIDE = None


def use_library(path):
    """Developer hook for A/B runs of two builds (tools/): load `path` instead of lib/libprimalcr.so.  Call before anything
    else of the package; the product path reads no environment variable (include/primalcr.h)."""
    global _LIB_OVERRIDE, _lib
    _LIB_OVERRIDE, _lib = path, None


def lib_path() -> str:
    return _LIB_OVERRIDE or os.path.join(_HERE, "lib", "libprimalcr.so")


class PcrError(RuntimeError):
    pass


class Parameter(C.Structure):
    """class parameter, pmf.h:9-49 (fields the PrimalCR/PrimalCR++ path reads) + device extensions."""
    _fields_ = [("solver_type", C.c_int), ("k", C.c_int), ("threads", C.c_int), ("maxiter", C.c_int),
                ("lambda_", C.c_double), ("do_predict", C.c_int), ("verbose", C.c_int), ("stepsize", C.c_double),
                ("ndcg_k", C.c_int), ("precision", C.c_int), ("device", C.c_int), ("cg_max_iter", C.c_int),
                ("cg_tol", C.c_double)]

    def __init__(self, **kw):
        super().__init__()
        lib().pcr_params_default(C.byref(self))
        for k, v in kw.items():
            if k == "lambda":
                k = "lambda_"
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class IterStats(C.Structure):
    _fields_ = [("obj", C.c_double), ("train_err", C.c_double), ("train_ndcg", C.c_double), ("test_err", C.c_double),
                ("test_ndcg", C.c_double), ("seconds", C.c_double), ("cg_v", C.c_int64), ("ls_v", C.c_int64),
                ("cg_u", C.c_int64), ("ls_u", C.c_int64)]


_LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
_lib = None
_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def lib():
    """Load libprimalcr.so; fail loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise PcrError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(make -C primalcr_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(path)
    vp, ci, cd, i64 = C.c_void_p, C.c_int, C.c_double, C.c_int64
    L.pcr_last_error.restype = C.c_char_p
    L.pcr_version.restype = C.c_char_p
    L.pcr_params_default.argtypes = [C.POINTER(Parameter)]
    L.pcr_initial.argtypes = [_dp, i64, i64]
    L.pcr_initial_rows.argtypes = [_dp, i64, i64, i64, i64]
    L.pcr_dataset_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.pcr_dataset_load_mt.argtypes = [C.c_char_p, ci, C.POINTER(vp)]
    L.pcr_dataset_load_cached.argtypes = [C.c_char_p, ci, C.c_char_p, C.POINTER(vp)]
    L.pcr_dataset_load_cache.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.pcr_dataset_save_cache.argtypes = [vp, C.c_char_p]
    L.pcr_dataset_from_triplets.argtypes = [i64, i64, i64, vp, vp, vp, i64, vp, vp, vp, C.POINTER(vp)]
    L.pcr_dataset_from_csr.argtypes = [i64, i64, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.pcr_dataset_free.argtypes = [vp]
    L.pcr_tune.argtypes = [C.c_char_p, C.c_char_p]
    L.pcr_dataset_dims.argtypes = [vp] + [C.POINTER(i64)] * 4
    L.pcr_dataset_csr.argtypes = [vp, ci, vp, vp, vp]
    L.pcr_dataset_count_pairs.argtypes = [vp, ci]
    L.pcr_dataset_count_pairs.restype = i64
    L.pcr_model_save.argtypes = [C.c_char_p, _dp, i64, _dp, i64, i64]
    L.pcr_model_load.argtypes = [C.c_char_p, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), vp, vp]
    L.pcr_partition_users.argtypes = [_lp, i64, ci, _lp]
    L.pcr_solver_create.argtypes = [vp, C.POINTER(Parameter), ci, ci, C.POINTER(vp)]
    L.pcr_solver_create_shard.argtypes = [vp, C.POINTER(Parameter), ci, ci, i64, i64, C.POINTER(vp)]
    L.pcr_solver_destroy.argtypes = [vp]
    L.pcr_comm_unique_id.argtypes = [vp]
    L.pcr_solver_comm_init.argtypes = [vp, vp]
    L.pcr_solver_comm_init_p2p.argtypes = [vp, C.c_char_p]
    L.pcr_solver_comm_nranks.argtypes = [vp]
    L.pcr_solver_counter.argtypes = [vp, C.c_char_p, C.POINTER(cd)]
    L.pcr_solver_setup_phase.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(cd)]
    L.pcr_solver_ustep_classes.argtypes = [vp, C.c_char_p, i64]
    L.pcr_solver_set_local_only.argtypes = [vp, ci]
    L.pcr_solver_shard.argtypes = [vp] + [C.POINTER(i64)] * 3
    L.pcr_solver_set_factors.argtypes = [vp, vp, vp]
    L.pcr_solver_get_factors.argtypes = [vp, vp, vp]
    L.pcr_solver_set_factors_local.argtypes = [vp, vp, vp]
    L.pcr_solver_get_factors_local.argtypes = [vp, vp, vp]
    L.pcr_comp_m.argtypes = [vp, vp]
    L.pcr_objective.argtypes = [vp, C.POINTER(cd)]
    L.pcr_obtain_g.argtypes = [vp, _dp]
    L.pcr_compute_Ha.argtypes = [vp, _dp, _dp]
    L.pcr_solve_delta.argtypes = [vp, _dp, _dp, C.POINTER(ci)]
    L.pcr_update_V.argtypes = [vp, C.POINTER(cd), C.POINTER(ci)]
    L.pcr_update_U.argtypes = [vp, C.POINTER(cd), C.POINTER(i64)]
    L.pcr_evaluate.argtypes = [vp, ci, ci, C.POINTER(cd), C.POINTER(cd)]
    L.pcr_train.argtypes = [vp, vp, vp, C.POINTER(IterStats)]
    L.pcr_iterate.argtypes = [vp, ci, C.POINTER(IterStats)]
    L.pcr_predict.argtypes = [_dp, i64, _dp, i64, i64, i64, _ip, _ip, _dp, ci]
    L.pcr_profile_enable.argtypes = [vp, ci]
    L.pcr_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(cd), C.POINTER(i64)]
    L.pcr_profile_scope.argtypes = [vp, C.c_char_p, C.POINTER(i64), C.POINTER(i64)]
    L.pcr_profile_launches.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    L.pcr_profile_reset.argtypes = [vp]
    L.pcr_profile_list.argtypes = [vp, C.c_char_p, i64]
    L.pcr_solver_sync.argtypes = [vp]
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise PcrError(f"libprimalcr error {rc}: {lib().pcr_last_error().decode()}")


def tune(key, value=None):
    """pcr_tune(): a process-wide launch knob read by the next Solver (value None removes it)."""
    _chk(lib().pcr_tune(key.encode(), None if value is None else str(value).encode()))


class tuned:
    """with tuned(spmm_tiles=16, lanes=1): ... -- knobs set for the block, removed afterwards."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            tune(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            tune(k, None)
        return False


def initial(n, k):
    """util.cpp:80-93 initial()."""
    X = np.empty((n, k), np.float64)
    _chk(lib().pcr_initial(X, n, k))
    return X


def initial_rows(n, k, row0, nrows):
    """Rows [row0, row0 + nrows) of initial(n, k) (the stream is sequential: the rows before are generated and dropped)."""
    X = np.empty((nrows, k), np.float64)
    _chk(lib().pcr_initial_rows(X, n, k, row0, nrows))
    return X


def model_save(path, U, V):
    U = np.ascontiguousarray(U, np.float64); V = np.ascontiguousarray(V, np.float64)
    _chk(lib().pcr_model_save(os.fsencode(path), U, U.shape[0], V, V.shape[0], U.shape[1]))


def model_load(path):
    d1, d2, k = C.c_int64(), C.c_int64(), C.c_int64()
    _chk(lib().pcr_model_load(os.fsencode(path), d1, d2, k, None, None))
    U = np.empty((d1.value, k.value)); V = np.empty((d2.value, k.value))
    _chk(lib().pcr_model_load(os.fsencode(path), d1, d2, k, U.ctypes.data, V.ctypes.data))
    return U, V


def partition_users(index, nparts):
    index = np.ascontiguousarray(index, np.int64)
    b = np.empty(nparts + 1, np.int64)
    _chk(lib().pcr_partition_users(index, index.shape[0] - 1, nparts, b))
    return b


def predict(U, V, user, item, device=0):
    """pmf-predict.cpp:56-64 on the GPU (0-based pairs)."""
    U = np.ascontiguousarray(U, np.float64); V = np.ascontiguousarray(V, np.float64)
    user = np.ascontiguousarray(user, np.int32); item = np.ascontiguousarray(item, np.int32)
    out = np.empty(user.shape[0], np.float64)
    _chk(lib().pcr_predict(U, U.shape[0], V, V.shape[0], U.shape[1], user.shape[0], user, item, out, device))
    return out


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _chk(lib().pcr_comm_unique_id(buf))
    return buf.raw


class Dataset:
    """load() + convert(): util.cpp:6-25, util.cpp:219-274."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path, cache=None, threads=0):
        """cache: path of the binary side-car (read if it matches the text files, else rebuilt)."""
        h = C.c_void_p()
        if cache is None and threads > 0:
            _chk(lib().pcr_dataset_load_mt(os.fsencode(path), threads, C.byref(h)))
        elif cache is None:
            _chk(lib().pcr_dataset_load(os.fsencode(path), C.byref(h)))
        else:
            _chk(lib().pcr_dataset_load_cached(os.fsencode(path), threads, os.fsencode(cache), C.byref(h)))
        return cls(h)

    @classmethod
    def load_cache(cls, cache):
        h = C.c_void_p()
        _chk(lib().pcr_dataset_load_cache(os.fsencode(cache), C.byref(h)))
        return cls(h)

    def save_cache(self, cache):
        _chk(lib().pcr_dataset_save_cache(self._h, os.fsencode(cache)))

    @classmethod
    def from_triplets(cls, d1, d2, user, item, val, tuser=None, titem=None, tval=None):
        user = np.ascontiguousarray(user, np.int32); item = np.ascontiguousarray(item, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        tn = 0 if tuser is None else len(tuser)
        if tn:
            tuser = np.ascontiguousarray(tuser, np.int32); titem = np.ascontiguousarray(titem, np.int32)
            tval = np.ascontiguousarray(tval, np.float64)
            targs = (tuser.ctypes.data, titem.ctypes.data, tval.ctypes.data)
        else:
            targs = (None, None, None)
        h = C.c_void_p()
        _chk(lib().pcr_dataset_from_triplets(d1, d2, len(user), user.ctypes.data, item.ctypes.data, val.ctypes.data,
                                             tn, *targs, C.byref(h)))
        return cls(h)

    @classmethod
    def from_csr(cls, d1, d2, index, item, val, tindex=None, titem=None, tval=None):
        """From arrays already in the SparseMat layout (index int64[d1+1], item int32 ascending per user, val float64)."""
        index = np.ascontiguousarray(index, np.int64); item = np.ascontiguousarray(item, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        targs = (None, None, None)
        if tindex is not None:
            tindex = np.ascontiguousarray(tindex, np.int64); titem = np.ascontiguousarray(titem, np.int32)
            tval = np.ascontiguousarray(tval, np.float64)
            targs = (tindex.ctypes.data, titem.ctypes.data, tval.ctypes.data)
        h = C.c_void_p()
        _chk(lib().pcr_dataset_from_csr(d1, d2, index.ctypes.data, item.ctypes.data, val.ctypes.data, *targs, C.byref(h)))
        return cls(h)

    @classmethod
    def from_ratings(cls, R):
        if hasattr(R, "index"):                       # synth.CsrRatings (the C++ generator): no triplet round trip
            return cls.from_csr(R.d1, R.d2, R.index, R.item, R.val, R.tindex, R.titem, R.tval)
        return cls.from_triplets(R.d1, R.d2, R.user, R.item, R.val, R.tuser, R.titem, R.tval)

    def dims(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _chk(lib().pcr_dataset_dims(self._h, a, b, c, d))
        return a.value, b.value, c.value, d.value

    def csr(self, which=0):
        d1, d2, nnz, tnnz = self.dims()
        n = nnz if which == 0 else tnnz
        idx = np.empty(d1 + 1, np.int64); item = np.empty(max(n, 1), np.int64); val = np.empty(max(n, 1), np.float64)
        _chk(lib().pcr_dataset_csr(self._h, which, idx.ctypes.data, item.ctypes.data, val.ctypes.data))
        return idx, item[:n], val[:n]

    def count_pairs(self, solver_type=PCR_SOLVER_PCRPP):
        return lib().pcr_dataset_count_pairs(self._h, solver_type)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pcr_dataset_free(self._h)
            self._h = None


class Solver:
    """Device solver: the reference's pcrpp()/pcr() and their building blocks on one MI355X."""

    def __init__(self, ds: Dataset, param: Parameter, rank=0, nranks=1, shard=None):
        """shard = (first_user, d1_total): ds holds ONLY this rank's users (pcr_solver_create_shard)."""
        self.ds, self.param, self.rank, self.nranks = ds, param, rank, nranks
        self.d1, self.d2, self.nnz, self.tnnz = ds.dims()
        self.k = param.k
        h = C.c_void_p()
        if shard is None:
            _chk(lib().pcr_solver_create(ds._h, C.byref(param), rank, nranks, C.byref(h)))
        else:
            _chk(lib().pcr_solver_create_shard(ds._h, C.byref(param), rank, nranks, int(shard[0]), int(shard[1]), C.byref(h)))
            self.d1 = int(shard[1])
        self._h = h
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(lib().pcr_solver_shard(h, a, b, c))
        self.first_user, self.n_users, self.nnz_local = a.value, b.value, c.value

    def comm_init(self, uid: bytes):
        buf = C.create_string_buffer(uid, 128)
        _chk(lib().pcr_solver_comm_init(self._h, buf))

    def comm_init_p2p(self, shm_name: str):
        """Direct peer-to-peer exchange (every rank of the node calls it with the same '/name')."""
        _chk(lib().pcr_solver_comm_init_p2p(self._h, shm_name.encode()))

    def counter(self, name):
        v = C.c_double()
        _chk(lib().pcr_solver_counter(self._h, name.encode(), v))
        return v.value

    def setup_phases(self):
        """[(phase, ms)] of pcr_solver_create, in order."""
        out, i = [], 0
        name, ms = C.c_char_p(), C.c_double()
        while lib().pcr_solver_setup_phase(self._h, i, C.byref(name), C.byref(ms)) == 0:
            out.append((name.value.decode(), ms.value)); i += 1
        return out

    def ustep_classes(self):
        """Profile slot names of the U step's length classes on this rank."""
        buf = C.create_string_buffer(4096)
        _chk(lib().pcr_solver_ustep_classes(self._h, buf, 4096))
        return [n for n in buf.value.decode().split(",") if n]

    def class_row_gathers(self):
        """{U-step class slot: rows of V it has gathered since the solver was created} (needs tuned(count_rows=1))."""
        return {n: self.counter("ustep_row_gathers/" + n) for n in self.ustep_classes()}

    def comm_nranks(self):
        return lib().pcr_solver_comm_nranks(self._h)

    def set_local_only(self, on=True):
        _chk(lib().pcr_solver_set_local_only(self._h, int(on)))

    def set_factors(self, U=None, V=None):
        U = None if U is None else np.ascontiguousarray(U, np.float64)
        V = None if V is None else np.ascontiguousarray(V, np.float64)
        _chk(lib().pcr_solver_set_factors(self._h, None if U is None else U.ctypes.data,
                                          None if V is None else V.ctypes.data))

    def get_factors(self):
        U = np.zeros((self.d1, self.k)); V = np.zeros((self.d2, self.k))
        _chk(lib().pcr_solver_get_factors(self._h, U.ctypes.data, V.ctypes.data))
        return U, V

    def set_factors_local(self, U_local=None, V=None):
        """U_local: this rank's n_users x k rows only."""
        U = None if U_local is None else np.ascontiguousarray(U_local, np.float64)
        V = None if V is None else np.ascontiguousarray(V, np.float64)
        if U is not None and U.shape != (self.n_users, self.k):
            raise ValueError(f"U_local must be {self.n_users} x {self.k}")
        _chk(lib().pcr_solver_set_factors_local(self._h, None if U is None else U.ctypes.data, None if V is None else V.ctypes.data))

    def get_factors_local(self):
        U = np.zeros((self.n_users, self.k)); V = np.zeros((self.d2, self.k))
        _chk(lib().pcr_solver_get_factors_local(self._h, U.ctypes.data, V.ctypes.data))
        return U, V

    def comp_m(self, want=True):
        m = np.empty(max(self.nnz_local, 1)) if want else None
        _chk(lib().pcr_comp_m(self._h, None if m is None else m.ctypes.data))
        return None if m is None else m[:self.nnz_local]

    def objective(self):
        o = C.c_double()
        _chk(lib().pcr_objective(self._h, o))
        return o.value

    def obtain_g(self):
        g = np.empty((self.d2, self.k))
        _chk(lib().pcr_obtain_g(self._h, g))
        return g

    def compute_Ha(self, a):
        a = np.ascontiguousarray(a, np.float64)
        Ha = np.empty((self.d2, self.k))
        _chk(lib().pcr_compute_Ha(self._h, a, Ha))
        return Ha

    def solve_delta(self, g):
        g = np.ascontiguousarray(g, np.float64)
        d = np.empty((self.d2, self.k)); it = C.c_int()
        _chk(lib().pcr_solve_delta(self._h, g, d, it))
        return d, it.value

    def update_V(self):
        o = C.c_double(); info = (C.c_int * 3)()
        _chk(lib().pcr_update_V(self._h, o, info))
        return o.value, dict(cg=info[0], ls=info[1], accepted=info[2])

    def update_U(self):
        o = C.c_double(); info = (C.c_int64 * 2)()
        _chk(lib().pcr_update_U(self._h, o, info))
        return o.value, dict(cg=info[0], ls=info[1])

    def evaluate(self, which=0, ndcg_k=10):
        e, n = C.c_double(), C.c_double()
        _chk(lib().pcr_evaluate(self._h, which, ndcg_k, e, n))
        return e.value, n.value

    def train(self, log=None):
        """pcrpp()/pcr(): returns (per-iteration records, log lines)."""
        hist = (IterStats * (self.param.maxiter + 1))()
        lines = []

        def _cb(_ctx, s):
            lines.append(s.decode())
            if log:
                log(s.decode())
        cb = _LOG_FN(_cb)
        _chk(lib().pcr_train(self._h, C.cast(cb, C.c_void_p), None, hist))
        recs = [{k: getattr(h, k) for k, _ in IterStats._fields_} for h in hist]
        return recs, lines

    def iterate(self, n):
        """n outer iterations (V step + U step, no evaluation) with one host round trip each; returns their records."""
        hist = (IterStats * max(n, 1))()
        _chk(lib().pcr_iterate(self._h, n, hist))
        return [{k: getattr(h, k) for k, _ in IterStats._fields_} for h in hist[:n]]

    def profile(self, on=True, period=1):
        """Per-kernel HIP-event timing; period > 1 times every period-th launch of each kernel."""
        _chk(lib().pcr_profile_enable(self._h, (period if period > 1 else 1) if on else 0))

    def profile_reset(self):
        _chk(lib().pcr_profile_reset(self._h))

    def profile_get(self, name):
        ms, n = C.c_double(), C.c_int64()
        _chk(lib().pcr_profile_get(self._h, name.encode(), ms, n))
        return ms.value, n.value

    def profile_launches(self, name):
        """Launches of the slot since the last reset, timed or not."""
        n = C.c_int64()
        _chk(lib().pcr_profile_launches(self._h, name.encode(), n))
        return n.value

    def profile_scope(self, name):
        """(ratings, users) one launch of the slot covers on this rank."""
        a, b = C.c_int64(), C.c_int64()
        _chk(lib().pcr_profile_scope(self._h, name.encode(), a, b))
        return a.value, b.value

    def profile_all(self):
        """{slot name: (total ms, launches)} of every kernel timed so far."""
        buf = C.create_string_buffer(4096)
        _chk(lib().pcr_profile_list(self._h, buf, 4096))
        names = [n for n in buf.value.decode().split(",") if n]
        return {n: self.profile_get(n) for n in names}

    def sync(self):
        _chk(lib().pcr_solver_sync(self._h))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pcr_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
