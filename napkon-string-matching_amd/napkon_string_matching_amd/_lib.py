"""ctypes binding of ``libnsm_hip.so`` (C ABI: include/nsm_hip.h).  Fails loudly."""
import ctypes
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("NSM_HIP_LIBRARY", _HERE.parent / "csrc" / "libnsm_hip.so"))

ABI_VERSION = 5
FLAG_PRUNE = 1
FLAG_WAVE_WIDE = 2  # nsm_indel_levels_grid without the block-cooperative parking (A/B runs, tests)
FLAG_INDEX, FLAG_NO_INDEX = 4, 8  # nsm_jaccard_raw_grid: force / forbid the inverted-index kernels
FLAG_TILE_INDEX = 64  # with FLAG_INDEX: the per-tile LDS index even when the right table carries a global one
FLAG_RAW_SCORE = 32  # nsm_*_any_grid: the RAW plugin call instead of compare_terms
FLAG_ONE_STAGE = 1024  # nsm_indel_raw_grid: the 32-bucket histogram test for every pair (no 16-bucket first stage)
FLAG_SPLIT, FLAG_TILE, FLAG_PROBE = 128, 256, 512  # nsm_indel_levels_grid, one-word strings: force the split path / the tile kernel; scan only
FLAG_PARK = 16  # nsm_indel_levels_grid, strings > 64 code units: the round-2 park kernel instead of the shared-tile kernel
BUILD_PARTITION, BUILD_VALIDATE, BUILD_SORT = 1, 2, 4
CAT_NONE, CAT_INTERSECT, CAT_INTERSECT_OR_BOTH_EMPTY = 0, 1, 2

c_i32p = ctypes.c_void_p  # device pointers travel as integers
c_u64 = ctypes.c_uint64


class NsmHit(ctypes.Structure):
    _fields_ = [("score", ctypes.c_double), ("i", ctypes.c_int32), ("j", ctypes.c_int32)]


class NsmSetTable(ctypes.Structure):
    _fields_ = [
        ("ids", ctypes.c_void_p),
        ("cnt", ctypes.c_void_p),
        ("sig", ctypes.c_void_p),
        ("sig2", ctypes.c_void_p),
        ("orig", ctypes.c_void_p),
        ("size_start", ctypes.c_void_p),
        ("nlev", ctypes.c_void_p),
        ("plen", ctypes.c_void_p),
        ("cat", ctypes.c_void_p),
        ("filt", ctypes.c_void_p),
        ("seg", ctypes.c_void_p),
        ("seg_start", ctypes.c_void_p),
        ("n", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("max_levels", ctypes.c_int32),
        ("vocab", ctypes.c_int32),
        ("post", ctypes.c_void_p),
        ("post_start", ctypes.c_void_p),
        ("post_sq", ctypes.c_uint64 * 5),
        ("post_row_bits", ctypes.c_int32),
        ("post_format", ctypes.c_int32),
    ]


class NsmStrTable(ctypes.Structure):
    _fields_ = [
        ("codes", ctypes.c_void_p),
        ("len", ctypes.c_void_p),
        ("orig", ctypes.c_void_p),
        ("len_start", ctypes.c_void_p),
        ("hist", ctypes.c_void_p),
        ("n", ctypes.c_int32),
        ("stride", ctypes.c_int32),
        ("alphabet", ctypes.c_int32),
        ("hist16", ctypes.c_void_p),
    ]


class NsmAnyStrings(ctypes.Structure):
    _fields_ = [("codes", ctypes.c_void_p), ("offset", ctypes.c_void_p), ("n_rows", ctypes.c_int32),
                ("alphabet", ctypes.c_int32), ("max_len", ctypes.c_int32)]


class NsmAnyItems(ctypes.Structure):
    _fields_ = [("first", ctypes.c_void_p), ("nlev", ctypes.c_void_p), ("orig", ctypes.c_void_p), ("cat", ctypes.c_void_p),
                ("n", ctypes.c_int32)]


class NsmAnySets(ctypes.Structure):
    _fields_ = [("ids", ctypes.c_void_p), ("lv", ctypes.c_void_p), ("offset", ctypes.c_void_p), ("nlev", ctypes.c_void_p),
                ("plen", ctypes.c_void_p), ("orig", ctypes.c_void_p), ("cat", ctypes.c_void_p), ("n", ctypes.c_int32),
                ("max_levels", ctypes.c_int32), ("max_ids", ctypes.c_int32), ("first", ctypes.c_void_p)]


class NsmLevelItems(ctypes.Structure):
    _fields_ = [
        ("first", ctypes.c_void_p),
        ("nlev", ctypes.c_void_p),
        ("orig", ctypes.c_void_p),
        ("cat", ctypes.c_void_p),
        ("seg", ctypes.c_void_p),
        ("seg_start", ctypes.c_void_p),
        ("n", ctypes.c_int32),
    ]


EXPORTS = (
    "nsm_abi_version",
    "nsm_last_error",
    "nsm_jaccard_raw_grid",
    "nsm_jaccard_levels_grid",
    "nsm_indel_raw_grid",
    "nsm_indel_levels_grid",
    "nsm_indel_levels_workspace_bytes",
    "nsm_release",
    "nsm_release_all",
    "nsm_sort_hits",
    "nsm_build_set_table",
    "nsm_build_str_table",
    "nsm_build_level_items",
    "nsm_indel_any_grid",
    "nsm_jaccard_any_grid",
)

_lib = None


class NsmLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the in-tree HIP library once.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NsmLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the match loop."
        )
    lib = ctypes.CDLL(str(LIB_PATH))
    lib.nsm_abi_version.restype = ctypes.c_int
    lib.nsm_last_error.restype = ctypes.c_char_p
    if lib.nsm_abi_version() != ABI_VERSION:
        raise NsmLibraryError(f"{LIB_PATH}: ABI {lib.nsm_abi_version()} != expected {ABI_VERSION}")
    P = ctypes.POINTER
    grid_tail = [ctypes.c_void_p, c_u64, ctypes.c_void_p, ctypes.c_void_p]  # hits, capacity, hit_count, stream
    lib.nsm_jaccard_raw_grid.argtypes = [P(NsmSetTable), P(NsmSetTable), ctypes.c_double, ctypes.c_uint32] + grid_tail
    lib.nsm_jaccard_levels_grid.argtypes = [
        P(NsmSetTable), P(NsmSetTable), ctypes.c_double, ctypes.c_int32, ctypes.c_uint32] + grid_tail
    lib.nsm_indel_raw_grid.argtypes = [P(NsmStrTable), P(NsmStrTable), ctypes.c_double, ctypes.c_uint32] + grid_tail
    # hits, capacity, hit_count, workspace, workspace_bytes, expected_survivors, stream
    lib.nsm_indel_levels_grid.argtypes = [
        P(NsmLevelItems), P(NsmStrTable), P(NsmLevelItems), P(NsmStrTable),
        ctypes.c_double, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p, c_u64, ctypes.c_void_p, ctypes.c_void_p, c_u64,
        ctypes.c_double, ctypes.c_void_p]
    lib.nsm_indel_levels_workspace_bytes.argtypes = [
        P(NsmLevelItems), P(NsmStrTable), P(NsmLevelItems), P(NsmStrTable), ctypes.c_double, ctypes.c_uint32, ctypes.c_double]
    lib.nsm_release.argtypes = [ctypes.c_void_p]
    lib.nsm_release_all.argtypes = []
    lib.nsm_indel_any_grid.argtypes = [
        P(NsmAnyItems), P(NsmAnyStrings), P(NsmAnyItems), P(NsmAnyStrings), ctypes.c_double, ctypes.c_int32,
        ctypes.c_uint32] + grid_tail
    lib.nsm_jaccard_any_grid.argtypes = [P(NsmAnySets), P(NsmAnySets), ctypes.c_double, ctypes.c_int32, ctypes.c_uint32] + grid_tail
    # hits, scratch, capacity, hit_count, n_hint, id_limit, stream
    lib.nsm_sort_hits.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_u64, ctypes.c_void_p, c_u64, ctypes.c_uint32,
                                  ctypes.c_void_p]
    vp, i32, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32
    lib.nsm_build_set_table.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, u32, P(NsmSetTable), vp]
    lib.nsm_build_str_table.argtypes = [vp, vp, vp, i32, u32, P(NsmStrTable), vp]
    lib.nsm_build_level_items.argtypes = [vp, vp, vp, vp, i32, i32, u32, P(NsmLevelItems), vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("nsm_last_error", "nsm_indel_levels_workspace_bytes"):
            fn.restype = ctypes.c_int
    lib.nsm_indel_levels_workspace_bytes.restype = c_u64
    lib.nsm_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().nsm_last_error().decode("utf-8", "replace")
        if status == 10002:
            raise NotImplementedError(f"{what}: {msg}")
        raise NsmLibraryError(f"{what} failed with status {status}: {msg}")
