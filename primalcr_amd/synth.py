"""Shape-matched synthetic rating sets.

The reference checkout ships no training files (``.MISSING_LARGE_BLOBS``), so every
BASELINE config runs on build-owned synthetic data of the same shape: per-user
counts are lognormal, items are drawn without replacement, ratings come from a
rank-8 ground-truth model cut at quantiles that reproduce the 1-5 star shares of
``ml1m/test.ratings`` (3047/5541/14122/21259/16431 of 60 400), plus ``n_test``
held-out ratings per user.  Deterministic for a given seed.

Files are written in the reference's input format (``util.cpp:6-25``,
``util.h:118-131``): ``meta`` + ``user item rating`` 1-based text, user-sorted,
items ascending (the test file MUST be user-sorted, ``util.cpp:259-261``).
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

import numpy as np

SEED = 20261004
LEVEL_SHARES = np.array([3047, 5541, 14122, 21259, 16431], dtype=np.float64) / 60400.0

SHAPES = {
    # name: (d1, d2, nnz, mu, sigma, real_valued, n_test)
    "tiny": (60, 40, 600, 2.0, 0.6, False, 3),
    "small": (500, 300, 20000, 3.4, 0.8, False, 5),
    "ml1m": (6040, 3952, 939809, 4.4, 1.0, False, 10),
    "toy": (6040, 3952, 900189, 4.4, 1.0, True, 10),
    "netflix": (480189, 17770, 100000000, 4.6, 1.2, False, 10),
    "yahoo": (1800000, 136000, 700000000, 4.9, 1.3, False, 10),
}


@dataclass
class Ratings:
    """Triplets, 0-based, user-sorted with items ascending inside a user."""
    d1: int
    d2: int
    user: np.ndarray   # int32
    item: np.ndarray   # int32
    val: np.ndarray    # float64
    tuser: np.ndarray
    titem: np.ndarray
    tval: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.user.shape[0])


def _user_counts(rng, d1, d2, nnz, mu, sigma, lo):
    hi = max(lo, d2 - 20)
    w = rng.lognormal(mu, sigma, size=d1)
    cnt = np.clip(np.rint(w * (nnz / w.sum())), lo, hi).astype(np.int64)
    # fix the total by nudging users that have slack, deterministically
    for _ in range(64):
        diff = int(nnz - cnt.sum())
        if diff == 0:
            break
        if diff > 0:
            cand = np.flatnonzero(cnt < hi)
        else:
            cand = np.flatnonzero(cnt > lo)
        if cand.size == 0:
            break
        step = min(abs(diff), cand.size)
        pick = cand[np.argsort(-cnt[cand], kind="stable")[:step]]
        cnt[pick] += 1 if diff > 0 else -1
    return cnt


def _sample_items(rng, cnt, d2, chunk_users=1 << 16):
    """Per-user uniform item sets without replacement, vectorised in chunks.
    Returns (user, item) sorted by (user, item)."""
    users_out, items_out = [], []
    d1 = cnt.shape[0]
    t0 = time.time()
    for u0 in range(0, d1, chunk_users):
        if d1 > 4 * chunk_users:       # large shapes take minutes: say so
            print(f"[synth] sampling users {u0}..{min(d1, u0 + chunk_users)} of {d1} ({time.time() - t0:.0f}s)", file=sys.stderr, flush=True)
        c = cnt[u0:u0 + chunk_users]
        need = c.copy()
        got_u = np.empty(0, np.int64)
        got_i = np.empty(0, np.int64)
        for _ in range(32):
            if need.sum() == 0:
                break
            draw = (need * 1.15).astype(np.int64) + 4
            draw[need == 0] = 0
            uu = np.repeat(np.arange(c.shape[0], dtype=np.int64), draw)
            ii = rng.integers(0, d2, size=uu.shape[0], dtype=np.int64)
            key = np.concatenate([got_u * d2 + got_i, uu * d2 + ii])
            fresh = np.concatenate([np.zeros(got_u.shape[0], bool), np.ones(uu.shape[0], bool)])
            # unique, preferring already-accepted entries
            order = np.lexsort((fresh, key))
            key, fresh = key[order], fresh[order]
            keep = np.ones(key.shape[0], bool)
            keep[1:] = key[1:] != key[:-1]
            key, fresh = key[keep], fresh[keep]
            ku = key // d2
            # trim each user's fresh entries to what it still needs (random subset)
            pri = rng.random(key.shape[0])
            pri[~fresh] = -1.0
            order = np.lexsort((pri, ku))
            key, ku = key[order], ku[order]
            start = np.zeros(c.shape[0] + 1, np.int64)
            np.add.at(start, ku + 1, 1)
            start = np.cumsum(start)
            rank = np.arange(key.shape[0], dtype=np.int64) - start[ku]
            sel = rank < c[ku]
            key, ku = key[sel], ku[sel]
            got_u, got_i = ku, key % d2
            have = np.bincount(got_u, minlength=c.shape[0])
            need = c - have
        order = np.lexsort((got_i, got_u))
        users_out.append((got_u[order] + u0).astype(np.int32))
        items_out.append(got_i[order].astype(np.int32))
    return np.concatenate(users_out), np.concatenate(items_out)


def _draw_sets(seed, d1, d2, nnz, mu, sigma, n_test, lo):
    """The part of generate() up to the ground-truth factors: (rng, cnt, cnt_all, user, item, Ug, Vg)."""
    rng = np.random.default_rng(seed)
    cnt = _user_counts(rng, d1, d2, nnz, mu, sigma, lo)
    cnt_all = np.minimum(cnt + n_test, d2)
    user, item = _sample_items(rng, cnt_all, d2)
    # rank-8 ground truth + noise
    k = 8
    Ug = rng.normal(0.0, np.sqrt(1.0 / k), size=(d1, k))
    Vg = rng.normal(0.0, 1.0, size=(d2, k))
    return rng, cnt, cnt_all, user, item, Ug, Vg


def generate(shape: str = "ml1m", seed: int = SEED, *, d1=None, d2=None, nnz=None,
             mu=None, sigma=None, real_valued=None, n_test=None, min_count=None, item_seed=None) -> Ratings:
    """Generate a synthetic rating set of a named shape (or override the fields).
    item_seed: take the items' ground-truth factors from the set of THAT seed (same shape) -- independent user blocks of one
    item catalogue (bench.py --gpus N: rank q's users are the set of seed SEED + q, the catalogue that of seed SEED)."""
    s = SHAPES[shape]
    d1 = s[0] if d1 is None else d1
    d2 = s[1] if d2 is None else d2
    nnz = s[2] if nnz is None else nnz
    mu = s[3] if mu is None else mu
    sigma = s[4] if sigma is None else sigma
    real_valued = s[5] if real_valued is None else real_valued
    n_test = s[6] if n_test is None else n_test
    lo = (10 if d2 >= 64 else 2) if min_count is None else min_count
    rng, cnt, cnt_all, user, item, Ug, Vg = _draw_sets(seed, d1, d2, nnz, mu, sigma, n_test, lo)
    if item_seed is not None and item_seed != seed:
        Vg = _draw_sets(item_seed, d1, d2, nnz, mu, sigma, n_test, lo)[6]
    score = np.empty(user.shape[0], np.float64)
    B = 1 << 22
    for a in range(0, user.shape[0], B):
        score[a:a + B] = np.einsum("ij,ij->i", Ug[user[a:a + B]], Vg[item[a:a + B]])
    score += rng.normal(0.0, 0.5, size=score.shape[0])
    if real_valued:
        val = score
    else:
        qs = np.cumsum(LEVEL_SHARES)[:-1]
        sample = score if score.shape[0] <= (1 << 22) else score[:: score.shape[0] // (1 << 22)]
        thr = np.quantile(sample, qs)
        val = 1.0 + np.searchsorted(thr, score, side="right").astype(np.float64)
    # hold out n_test ratings per user (random subset)
    start = np.zeros(d1 + 1, np.int64)
    np.add.at(start, user.astype(np.int64) + 1, 1)
    start = np.cumsum(start)
    pri = rng.random(user.shape[0])
    order = np.lexsort((pri, user))
    rank = np.empty(user.shape[0], np.int64)
    rank[order] = np.arange(user.shape[0], dtype=np.int64) - start[user[order]]
    ntest_u = np.minimum(n_test, np.maximum(cnt_all - cnt, 0))
    is_test = rank < ntest_u[user]
    tr = ~is_test
    return Ratings(d1, d2, user[tr], item[tr], np.ascontiguousarray(val[tr]),
                   user[is_test], item[is_test], np.ascontiguousarray(val[is_test]))


# ---------------------------------------------------------------------------------------
# C++ generator (primalcr_amd/csrc/pcr_synth.cpp -> lib/libpcrsynth.so): the same recipe with one counter-based random
# stream per user -- any user range on its own, multi-threaded, 100 M ratings in seconds.  The large shapes (netflix,
# yahoo) use it; the numpy generator above stays the definition of the ml1m / toy / small sets (BASELINE configs[0..2]).
# ---------------------------------------------------------------------------------------
class _SynthParams(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("d1", _C.c_int64), ("d2", _C.c_int64), ("nnz", _C.c_int64), ("mu", _C.c_double), ("sigma", _C.c_double),
                ("real_valued", _C.c_int32), ("n_test", _C.c_int32), ("min_count", _C.c_int32), ("reserved", _C.c_int32),
                ("seed", _C.c_uint64)]


@dataclass
class CsrRatings:
    """A user range [u0, u1) of a shape, as CSRs (users renumbered from 0; items ascending inside a user)."""
    d1: int
    d2: int
    index: np.ndarray    # int64[d1 + 1]
    item: np.ndarray     # int32
    val: np.ndarray      # float64
    tindex: np.ndarray
    titem: np.ndarray
    tval: np.ndarray
    u0: int = 0
    shape_d1: int = 0    # users of the whole shape

    @property
    def nnz(self) -> int:
        return int(self.index[-1])

    @property
    def user(self):
        return np.repeat(np.arange(self.d1, dtype=np.int32), np.diff(self.index))

    @property
    def tuser(self):
        return np.repeat(np.arange(self.d1, dtype=np.int32), np.diff(self.tindex))


_synth_lib = None
_synth_lib_override = None


def use_library(path):
    """Load `path` instead of lib/libpcrsynth.so (the sanitizer build of the generator, tests/test_sanitizers.py)."""
    global _synth_lib, _synth_lib_override
    _synth_lib, _synth_lib_override = None, path


def _slib():
    global _synth_lib
    if _synth_lib is None:
        import ctypes as C
        path = _synth_lib_override or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpcrsynth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with make -C primalcr_amd/csrc")
        L = C.CDLL(path)
        L.pcr_synth_counts.argtypes = [C.POINTER(_SynthParams), C.c_void_p, C.c_void_p]
        L.pcr_synth_fill.argtypes = [C.POINTER(_SynthParams), C.c_int64, C.c_int64] + [C.c_void_p] * 6 + [C.c_int]
        L.pcr_synth_write_text.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _synth_lib = L
    return _synth_lib


def generate_fast(shape: str = "netflix", seed: int = SEED, *, d1=None, d2=None, nnz=None, mu=None, sigma=None,
                  real_valued=None, n_test=None, min_count=None, users=None, threads=0, counts_only=False) -> CsrRatings:
    """The shape (or overrides) from the C++ generator; users=(u0, u1) generates only that user range of it;
    counts_only: just the per-user (training, held-out) rating counts of the whole shape -- what a rank needs to pick its
    nnz-balanced user range before it generates anything."""
    s = SHAPES[shape]
    d1 = s[0] if d1 is None else d1
    d2 = s[1] if d2 is None else d2
    nnz = s[2] if nnz is None else nnz
    P = _SynthParams(d1, d2, nnz, s[3] if mu is None else mu, s[4] if sigma is None else sigma,
                     int(s[5] if real_valued is None else real_valued), s[6] if n_test is None else n_test,
                     ((10 if d2 >= 64 else 2) if min_count is None else min_count), 0, seed)
    L = _slib()
    ctr, cte = np.empty(d1, np.int64), np.empty(d1, np.int64)
    if L.pcr_synth_counts(P, ctr.ctypes.data, cte.ctypes.data) != 0:
        raise RuntimeError("pcr_synth_counts failed")
    if counts_only:
        return ctr, cte
    u0, u1 = (0, d1) if users is None else users
    index = np.concatenate([[0], np.cumsum(ctr[u0:u1])]).astype(np.int64)
    tindex = np.concatenate([[0], np.cumsum(cte[u0:u1])]).astype(np.int64)
    item, val = np.empty(int(index[-1]), np.int32), np.empty(int(index[-1]), np.float64)
    titem, tval = np.empty(int(tindex[-1]), np.int32), np.empty(int(tindex[-1]), np.float64)
    if L.pcr_synth_fill(P, u0, u1, ctr.ctypes.data, cte.ctypes.data, item.ctypes.data, val.ctypes.data, titem.ctypes.data,
                        tval.ctypes.data, threads) != 0:
        raise RuntimeError("pcr_synth_fill failed")
    return CsrRatings(u1 - u0, d2, index, item, val, tindex, titem, tval, u0, d1)


def _write_ratings(path, user, item, val, real_valued):
    with open(path, "w") as f:
        B = 1 << 20
        for a in range(0, user.shape[0], B):
            u = user[a:a + B].astype(np.int64) + 1
            i = item[a:a + B].astype(np.int64) + 1
            v = val[a:a + B]
            if real_valued:
                lines = [f"{x} {y} {z:.17g}" for x, y, z in zip(u.tolist(), i.tolist(), v.tolist())]
            else:
                lines = [f"{x} {y} {int(z)}" for x, y, z in zip(u.tolist(), i.tolist(), v.tolist())]
            f.write("\n".join(lines))
            f.write("\n")


def write_dir_fast(r: CsrRatings, path: str, train_name="training.ratings", test_name="test.ratings", threads=0):
    """write_dir for a CsrRatings through the C++ writer (100 M ratings in seconds): same files, byte for byte."""
    os.makedirs(path, exist_ok=True)
    L = _slib()
    index, tindex = np.ascontiguousarray(r.index, np.int64), np.ascontiguousarray(r.tindex, np.int64)
    item, val = np.ascontiguousarray(r.item, np.int32), np.ascontiguousarray(r.val, np.float64)
    titem, tval = np.ascontiguousarray(r.titem, np.int32), np.ascontiguousarray(r.tval, np.float64)
    if L.pcr_synth_write_text(os.fsencode(os.path.join(path, train_name)), r.d1, 0, index.ctypes.data, item.ctypes.data,
                              val.ctypes.data, threads) != 0:
        raise OSError(f"could not write {path}/{train_name}")
    with open(os.path.join(path, "meta"), "w") as f:
        f.write(f"{r.d1} {r.d2}\n{int(index[-1])} {train_name}\n")
        if int(tindex[-1]) > 0:
            if L.pcr_synth_write_text(os.fsencode(os.path.join(path, test_name)), r.d1, 0, tindex.ctypes.data, titem.ctypes.data,
                                      tval.ctypes.data, threads) != 0:
                raise OSError(f"could not write {path}/{test_name}")
            f.write(f"{int(tindex[-1])} {test_name}\n")
    return path


def write_dir(r: Ratings, path: str, train_name="training.ratings", test_name="test.ratings"):
    """Write ``meta`` + rating files in the reference's data-dir format."""
    if hasattr(r, "index"):
        return write_dir_fast(r, path, train_name, test_name)
    os.makedirs(path, exist_ok=True)
    real_valued = bool(np.any(r.val != np.rint(r.val)))
    _write_ratings(os.path.join(path, train_name), r.user, r.item, r.val, real_valued)
    with open(os.path.join(path, "meta"), "w") as f:
        f.write(f"{r.d1} {r.d2}\n{r.nnz} {train_name}\n")
        if r.tuser.shape[0] > 0:
            _write_ratings(os.path.join(path, test_name), r.tuser, r.titem, r.tval, real_valued)
            f.write(f"{r.tuser.shape[0]} {test_name}\n")
    return path


def count_pairs(r: Ratings) -> int:
    """#Omega = #{(i,j,k): R_ij > R_ik} on lround-bucketed ratings: the number of
    ordered pairs the objective sums over (the metric's unit)."""
    lv = np.rint(r.val).astype(np.int64)
    lv -= lv.min()
    T = int(lv.max()) + 1
    hist = np.zeros((r.d1, T), np.int64)
    np.add.at(hist, (r.user.astype(np.int64), lv), 1)
    n = hist.sum(1)
    return int(((n * n - (hist * hist).sum(1)) // 2).sum())
