"""spaghettisearch_amd — MI355X-native ranking hot path of SpaghettiSearch.

Scope: Topic-Sensitive PageRank power iteration
(ranking/pagerank.go), TF-IDF weight/magnitude build (ranking/term_weighting.go)
and the cosine vector-space scorer + PageRank blend + top-k
(retrieval/main_retrieve.go, get_metadata.go:16-77, util.go:48-54), as
hand-written HIP kernels for gfx950 behind a C ABI (include/spaghetti_rank.h).

The compute lives in libspaghetti_rank.so; this package is the Python binding
and the host-side mirror of the reference's three entry points.  There is no
CPU fallback: importing the engine without the built library raises.
"""
from ._lib import LIB_PATH, SpaghettiError  # noqa: F401

__all__ = ["LIB_PATH", "SpaghettiError"]
