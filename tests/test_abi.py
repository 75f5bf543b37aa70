"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/spaghetti_rank.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "spaghetti_rank.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("ss_init", "ss_graph_create", "ss_pagerank_run", "ss_index_create", "ss_tfidf_build",
                 "ss_scorer_create", "ss_scorer_set_prior", "ss_score_topk", "ss_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from spaghettisearch_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"not exported: {missing}"
    # the Python binding covers exactly the declared set
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    header_abi = int(re.search(r"#define SS_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert _lib.load().ss_abi_version() == header_abi == _lib.ABI_VERSION == 4


def test_no_cpu_fallback_without_device():
    """On a box without a GPU ss_init must fail loudly (SS_ERR_NO_DEVICE), never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from spaghettisearch_amd import SpaghettiError, engine
    with pytest.raises(SpaghettiError) as ei:
        engine.Context(0)
    assert ei.value.code == 2


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spaghettisearch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), encoding="utf-8", errors="replace").read()
                code = "\n".join(l for l in src.splitlines() if not l.strip().startswith(("#", "//", "*", "/*")))
                assert "pyoracle" not in code and "liboracle" not in code and "oracle_np" not in code, f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: the header compiles as C99 (what cgo feeds to the C compiler) and a plain C
    program links against the library and calls it (no compute: ss_abi_version; ss_init may report no device)."""
    import subprocess
    from spaghettisearch_amd import _lib
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HEADER], check=True)
    src = tmp_path / "c_caller.c"
    src.write_text('#include <stdio.h>\n#include "spaghetti_rank.h"\n'
                   'int main(void) { ss_ctx* c = 0; int v = ss_abi_version(); int rc = ss_init(0, &c);\n'
                   '  printf("abi=%d init=%d\\n", v, rc); if (rc == SS_OK) ss_shutdown(c); return v == SS_ABI_VERSION ? 0 : 1; }\n')
    exe = tmp_path / "c_caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir,
                    "-lspaghetti_rank", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("abi=4 init=")
