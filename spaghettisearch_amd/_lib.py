"""ctypes binding of libspaghetti_rank.so (the C ABI in include/spaghetti_rank.h).

There is no CPU fallback: if the shared library is missing the import of the
product API fails loudly, and without a gfx950 device ``ss_init`` fails.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS_LIB_PATH") or os.path.join(_HERE, "libspaghetti_rank.so")   # SS_LIB_PATH: A/B builds

SS_OK = 0
SS_MAX_TOPK = 1024
SS_MAX_TOPICS = 64
SS_UNKNOWN_TERM = 0xFFFFFFFF

ERR_NAMES = {0: "SS_OK", 1: "SS_ERR_INVALID", 2: "SS_ERR_NO_DEVICE", 3: "SS_ERR_HIP", 4: "SS_ERR_OOM",
             5: "SS_ERR_UNSORTED", 6: "SS_ERR_STATE", 7: "SS_ERR_UNSUPPORTED", 8: "SS_ERR_COMM"}
SS_COMM_ID_BYTES = 128


class SpaghettiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class SsHit(C.Structure):
    _fields_ = [("doc", C.c_uint32), ("_pad", C.c_uint32), ("title", C.c_double), ("body", C.c_double),
                ("pagerank", C.c_double), ("final", C.c_double)]


class SsGraphInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_edges", C.c_uint64), ("n_nondangling", C.c_uint64),
                ("n_rows_local", C.c_uint64), ("n_edges_local", C.c_uint64), ("max_indeg", C.c_uint32),
                ("rank", C.c_int32), ("world", C.c_int32)]


_vp = C.c_void_p
_i32 = C.c_int32
_u64 = C.c_uint64
_f64 = C.c_double

# name -> (restype, argtypes); every symbol include/spaghetti_rank.h declares
PROTOTYPES = {
    "ss_abi_version": (_i32, []),
    "ss_init": (_i32, [_i32, C.POINTER(_vp)]),
    "ss_shutdown": (_i32, [_vp]),
    "ss_set_stream": (_i32, [_vp, _vp]),
    "ss_synchronize": (_i32, [_vp]),
    "ss_set_option": (_i32, [_vp, C.c_char_p, C.c_int64]),
    "ss_last_error": (C.c_char_p, [_vp]),
    "ss_comm_unique_id": (_i32, [_vp]),
    "ss_comm_init": (_i32, [_vp, _vp, _i32, _i32]),
    "ss_comm_split": (_i32, [_vp, _i32, _i32]),
    "ss_comm_destroy": (_i32, [_vp]),
    "ss_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "ss_comm_allreduce_u64": (_i32, [_vp, _vp, _u64]),
    "ss_comm_allgather": (_i32, [_vp, _vp, _vp, _u64]),
    "ss_graph_create": (_i32, [_vp, _u64, _u64, _vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "ss_graph_apply_delta": (_i32, [_vp, _u64, _u64, _vp, _vp, _vp]),
    "ss_graph_get_info": (_i32, [_vp, C.POINTER(SsGraphInfo)]),
    "ss_graph_destroy": (_i32, [_vp]),
    "ss_pagerank_run": (_i32, [_vp, _f64, _f64, _i32, _i32, _vp, _vp, _vp]),
    "ss_pr_create": (_i32, [_vp, _f64, _f64, _i32, _i32, _vp, C.POINTER(_vp)]),
    "ss_pr_destroy": (_i32, [_vp]),
    "ss_pr_set_teleport": (_i32, [_vp, _vp, _vp]),
    "ss_pr_begin": (_i32, [_vp]),
    "ss_pr_step": (_i32, [_vp, _i32]),
    "ss_pr_finalize": (_i32, [_vp]),
    "ss_pr_exchange_buffers": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_vp), C.POINTER(_u64)]),
    "ss_pr_exchange": (_i32, [_vp, _i32]),
    "ss_pagerank_run_sharded": (_i32, [_vp, _f64, _f64, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "ss_pagerank_run_group": (_i32, [_vp, _i32, _f64, _f64, _i32, _i32, _vp, _vp, _vp]),
    "ss_pr_status": (_i32, [_vp, _vp, C.POINTER(_i32), C.POINTER(_i32), _vp, _vp]),
    "ss_pr_read_local": (_i32, [_vp, _vp, _vp]),
    "ss_pr_read": (_i32, [_vp, _vp]),
    "ss_pr_probe": (_i32, [_vp, _i32, _i32, C.POINTER(C.c_float)]),
    "ss_index_create": (_i32, [_vp, _u64, _u64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "ss_index_destroy": (_i32, [_vp]),
    "ss_tfidf_build": (_i32, [_vp, _u64, _vp, _vp, _vp]),
    "ss_index_apply_delta": (_i32, [_vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp]),
    "ss_index_apply_delta_pos": (_i32, [_vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "ss_index_resize": (_i32, [_vp, _u64, _u64]),
    "ss_index_read_magnitudes": (_i32, [_vp, _u64, _vp, _vp]),
    "ss_index_refresh_magnitudes": (_i32, [_vp, _vp]),
    "ss_index_get_info": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "ss_index_read": (_i32, [_vp, _vp, _vp, _vp]),
    "ss_index_read_positions": (_i32, [_vp, _vp, _vp]),
    "ss_index_set_doc_freq": (_i32, [_vp, _vp]),
    "ss_index_set_weighted": (_i32, [_vp, _vp]),
    "ss_index_set_positions": (_i32, [_vp, _vp, _vp]),
    "ss_scorer_create": (_i32, [_vp, _vp, _vp, C.POINTER(_vp)]),
    "ss_scorer_destroy": (_i32, [_vp]),
    "ss_scorer_set_prior": (_i32, [_vp, _i32, _vp]),
    "ss_score_topk": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "ss_score_topk_phrase": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "ss_score_topk_submit": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ss_score_topk_collect": (_i32, [_vp, C.c_uint64, _vp, _vp]),
    "ss_merge_hits": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ss_last_kernel_ms": (_i32, [_vp, _i32, C.POINTER(C.c_float)]),
}

ABI_VERSION = 4   # SS_ABI_VERSION of include/spaghetti_rank.h this binding follows (tests/test_abi.py compares the two)
_lib = None


def load() -> C.CDLL:
    """Load the shared library and attach prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C spaghettisearch_amd/csrc). There is no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.
    # If this library pulled in /opt/rocm's copy first, a later `import torch` would bring up a
    # second runtime in the same process ("no ROCm-capable device"), and torch streams / device
    # pointers could not be shared with the library.  Importing torch first makes the dynamic
    # loader resolve our libamdhip64.so.7 dependency to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ss_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.ss_abi_version()}, this binding was written for {ABI_VERSION} "
                          "(include/spaghetti_rank.h: SS_ABI_VERSION) - rebuild the library")
    _lib = lib
    return lib


def check(rc: int, ctx=None) -> None:
    if rc != SS_OK:
        msg = load().ss_last_error(ctx)
        raise SpaghettiError(rc, msg.decode("utf-8", "replace") if msg else "")
