"""CPU check of the cgo shim (go/): there is no Go toolchain in this image, so nothing compiles go/spaghetti/spaghetti.go.
This test is the mechanical stand-in for the part of `go vet` that matters at the boundary: every `C.ss_*(...)` call in the
Go sources names a function that include/spaghetti_rank.h declares and passes exactly as many arguments as the prototype has
parameters; every `C.ss_*` type and every `C.SS_*` constant used exists in the header."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "spaghetti_rank.h")


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def prototypes():
    """name -> number of parameters, from the header"""
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"\b(?:int32_t|const\s+char\s*\*)\s*(ss_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else len(_split_top_level(params))
    return out


def _split_top_level(s):
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return [p for p in (x.strip() for x in parts)]


def go_calls():
    """(file, line, name, n_args) of every C.ss_*( call outside comments and the cgo preamble"""
    calls = []
    for path in sorted(glob.glob(os.path.join(ROOT, "go", "*", "*.go"))):
        src = open(path, encoding="utf-8").read()
        src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)       # block comments incl. the preamble
        src = "\n".join(re.sub(r"//.*$", "", line) for line in src.split("\n"))
        for m in re.finditer(r"\bC\.(ss_[a-z0-9_]+)\s*\(", src):
            i, depth = m.end(), 1
            while depth and i < len(src):
                if src[i] in "([{":
                    depth += 1
                elif src[i] in ")]}":
                    depth -= 1
                i += 1
            assert depth == 0, (path, m.group(1))
            args = src[m.end():i - 1].strip()
            n = 0 if args == "" else len(_split_top_level(args))
            calls.append((os.path.relpath(path, ROOT), src.count("\n", 0, m.start()) + 1, m.group(1), n))
    return calls


def test_every_cgo_call_matches_its_prototype():
    protos = prototypes()
    assert len(protos) >= 50 and protos["ss_score_topk"] == 9 and protos["ss_abi_version"] == 0 and protos["ss_last_error"] == 1
    calls = go_calls()
    assert len(calls) >= 35
    bad = []
    for path, line, name, n in calls:
        if name not in protos:
            if name in ("ss_ctx", "ss_graph", "ss_pr", "ss_index", "ss_scorer", "ss_hit", "ss_graph_info"):
                continue                                        # a conversion like (*C.ss_hit)(ptr): a type, not a call
            bad.append(f"{path}:{line}: {name} is not declared in spaghetti_rank.h")
        elif protos[name] != n:
            bad.append(f"{path}:{line}: {name} called with {n} arguments, the prototype has {protos[name]}")
    assert not bad, "\n".join(bad)
    # the shim binds the entry points of all three reference functions
    used = {name for _, _, name, _ in calls}
    for must in ("ss_init", "ss_graph_create", "ss_pagerank_run", "ss_index_create", "ss_tfidf_build", "ss_scorer_create",
                 "ss_scorer_set_prior", "ss_score_topk", "ss_score_topk_phrase", "ss_index_apply_delta_pos", "ss_graph_apply_delta",
                 "ss_comm_init", "ss_pagerank_run_sharded", "ss_merge_hits"):
        assert must in used, must


def test_cgo_types_and_constants_exist_in_the_header():
    text = _strip_c_comments(open(HEADER).read())
    types = set(re.findall(r"typedef\s+struct\s+(ss_[a-z_]+)", text))
    consts = set(re.findall(r"#define\s+(SS_[A-Z_0-9]+)", text)) | set(re.findall(r"\b(SS_[A-Z_0-9]+)\s*=", text))
    for path in sorted(glob.glob(os.path.join(ROOT, "go", "*", "*.go"))):
        src = open(path, encoding="utf-8").read()
        body = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        body = "\n".join(re.sub(r"//.*$", "", line) for line in body.split("\n"))
        for t in set(re.findall(r"\*?C\.(ss_[a-z_]+)\b(?!\s*\()", body)):
            assert t in types, (path, t)
        for c in set(re.findall(r"\bC\.(SS_[A-Z_0-9]+)\b", body)):
            assert c in consts, (path, c)
