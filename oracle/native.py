"""ctypes wrapper of the C oracle (oracle/c/nsm_oracle.c) -- TEST INFRASTRUCTURE.

Used where the pure-Python oracle is too slow (grids of 10^5 .. 10^8 pairs).  It is itself checked
against the Python oracle in tests/test_oracle_native.py.
"""
import ctypes
import subprocess
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libnsm_oracle.so"


class OracleHit(ctypes.Structure):
    _fields_ = [("score", ctypes.c_double), ("i", ctypes.c_int32), ("j", ctypes.c_int32)]


def build(force: bool = False) -> Path:
    src = HERE / "c" / "nsm_oracle.c"
    if force or not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE / "c")], check=True, capture_output=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(LIB))
        for name in ("oracle_jaccard_raw", "oracle_indel_raw", "oracle_levels"):
            getattr(_lib, name).restype = ctypes.c_longlong
    return _lib


def csr(rows: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(rows) + 1, dtype=np.int64)
    for k, r in enumerate(rows):
        off[k + 1] = off[k] + len(r)
    val = np.zeros(max(1, int(off[-1])), dtype=np.int32)
    for k, r in enumerate(rows):
        val[off[k]: off[k + 1]] = list(r)
    return val, off


def csr_from_padded(ids: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    valid = ids >= 0
    off = np.zeros(ids.shape[0] + 1, dtype=np.int64)
    np.cumsum(valid.sum(axis=1), out=off[1:])
    val = ids[valid].astype(np.int32)
    if val.size == 0:
        val = np.zeros(1, np.int32)
    return np.ascontiguousarray(val), off


def csr_from_codes(codes: np.ndarray, length: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    mask = np.arange(codes.shape[1])[None, :] < length[:, None]
    off = np.zeros(codes.shape[0] + 1, dtype=np.int64)
    np.cumsum(length, out=off[1:])
    val = codes[mask].astype(np.int32)
    if val.size == 0:
        val = np.zeros(1, np.int32)
    return np.ascontiguousarray(val), off


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _collect(n: int, buf) -> List[Tuple[float, int, int]]:
    hits = [(buf[k].score, buf[k].i, buf[k].j) for k in range(n)]
    hits.sort(key=lambda h: (-h[0], h[1], h[2]))
    return hits


def _run(fn, args_before, thr: float, cap: int):
    while True:
        buf = (OracleHit * cap)()
        n = fn(*args_before, ctypes.c_double(thr), buf, ctypes.c_longlong(cap))
        if n == -1:
            raise ZeroDivisionError("division by zero")
        if n == -2:
            raise IndexError("list index out of range")
        if n <= cap:
            return _collect(int(n), buf)
        cap = int(n)


def jaccard_raw(left: Tuple[np.ndarray, np.ndarray], right: Tuple[np.ndarray, np.ndarray], thr: float, cap: int = 1 << 16):
    (lv, lo), (rv, ro) = left, right
    return _run(lib().oracle_jaccard_raw,
                (_p(lv), _p(lo), ctypes.c_int(len(lo) - 1), _p(rv), _p(ro), ctypes.c_int(len(ro) - 1)), thr, cap)


def indel_raw(left: Tuple[np.ndarray, np.ndarray], right: Tuple[np.ndarray, np.ndarray], thr: float, cap: int = 1 << 16):
    (lv, lo), (rv, ro) = left, right
    return _run(lib().oracle_indel_raw,
                (_p(lv), _p(lo), ctypes.c_int(len(lo) - 1), _p(rv), _p(ro), ctypes.c_int(len(ro) - 1)), thr, cap)


def levels(use_indel: bool, left_items: Sequence[Sequence[Sequence[int]]], right_items: Sequence[Sequence[Sequence[int]]],
           thr: float, left_cat: Optional[np.ndarray] = None, right_cat: Optional[np.ndarray] = None, cat_mode: int = 0,
           cap: int = 1 << 16):
    """``*_items[k]`` = list of levels, each a list of ints (token ids, or code points for Indel)."""

    def flat(items):
        lev = np.zeros(len(items) + 1, dtype=np.int64)
        rows: List[Sequence[int]] = []
        for k, it in enumerate(items):
            rows.extend(it)
            lev[k + 1] = len(rows)
        val, off = csr(rows)
        return val, off, lev

    lv, lo, ll = flat(left_items)
    rv, ro, rl = flat(right_items)
    lc = None if left_cat is None else np.ascontiguousarray(left_cat, dtype=np.uint64)
    rc = None if right_cat is None else np.ascontiguousarray(right_cat, dtype=np.uint64)
    return _run(lib().oracle_levels,
                (ctypes.c_int(1 if use_indel else 0), _p(lv), _p(lo), _p(ll), ctypes.c_int(len(left_items)),
                 _p(rv), _p(ro), _p(rl), ctypes.c_int(len(right_items)), _p(lc), _p(rc), ctypes.c_int(cat_mode)),
                thr, cap)
