"""CPU check of the cgo shim (go/): there is no Go toolchain in this image, so nothing compiles go/spaghetti/spaghetti.go.
This test is the mechanical stand-in for the part of `go vet` that matters at the boundary: every `C.ss_*(...)` call in the
Go sources names a function that include/spaghetti_rank.h declares and passes exactly as many arguments as the prototype has
parameters; every `C.ss_*` type and every `C.SS_*` constant used exists in the header; and (round 5) every argument whose C type can
be read off the Go expression — a `C.int32_t(...)`-style conversion, one of the shim's slice-to-pointer helpers (`u32p`, `u64p`, `i32p`,
`f32p`, `f64p`), a `(*C.ss_hit)(...)` cast — has the type of the parameter it is passed for."""
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


def prototype_types():
    """name -> [normalised parameter type] from the header (names and const dropped, spaces squeezed: 'uint32_t*', 'int32_t', 'ss_hit*')"""
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"\b(?:int32_t|const\s+char\s*\*)\s*(ss_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = m.group(2).strip()
        types = []
        if params not in ("", "void"):
            for prm in _split_top_level(params):
                prm = re.sub(r"\bconst\b", " ", prm)
                mm = re.match(r"^\s*([A-Za-z_][A-Za-z0-9_]*(?:\s+[A-Za-z_][A-Za-z0-9_]*)*?)\s*((?:\*\s*(?:const\s*)?)*)\s*([A-Za-z_][A-Za-z0-9_]*)?\s*$", prm)
                assert mm, (m.group(1), prm)
                types.append(re.sub(r"\s+", " ", mm.group(1)).strip() + mm.group(2).replace(" ", ""))
        out[m.group(1)] = types
    return out


HELPERS = {"u32p": "uint32_t*", "u64p": "uint64_t*", "i32p": "int32_t*", "f32p": "float*", "f64p": "double*"}


def go_arg_type(expr):
    """the C type a Go argument expression certainly has, or None"""
    expr = expr.strip()
    m = re.match(r"^C\.([a-z0-9_]+)\(", expr)
    if m:
        return m.group(1)
    m = re.match(r"^([a-z0-9]+p)\(", expr)
    if m and m.group(1) in HELPERS:
        return HELPERS[m.group(1)]
    m = re.match(r"^\(\*C\.([a-z0-9_]+)\)\(", expr)
    if m:
        return m.group(1) + "*"
    return None


def test_cgo_argument_types_match_the_header():
    types = prototype_types()
    assert types["ss_score_topk"] == ["ss_scorer*", "int32_t", "uint32_t*", "uint32_t*", "int32_t*", "double*", "int32_t", "ss_hit*", "int32_t*"]
    checked, bad = 0, []
    for path in sorted(glob.glob(os.path.join(ROOT, "go", "*", "*.go"))):
        src = open(path, encoding="utf-8").read()
        src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
        src = "\n".join(re.sub(r"//.*$", "", line) for line in src.split("\n"))
        for m in re.finditer(r"\bC\.(ss_[a-z0-9_]+)\s*\(", src):
            name = m.group(1)
            if name not in types:
                continue
            i, depth = m.end(), 1
            while depth and i < len(src):
                depth += src[i] in "([{"
                depth -= src[i] in ")]}"
                i += 1
            args = _split_top_level(src[m.end():i - 1].strip()) if src[m.end():i - 1].strip() else []
            for pos, (arg, want) in enumerate(zip(args, types[name])):
                got = go_arg_type(arg)
                if got is None:
                    continue
                checked += 1
                if got != want and not (want == "void*" and got.endswith("*")):
                    bad.append(f"{os.path.relpath(path, ROOT)}:{src.count(chr(10), 0, m.start()) + 1}: {name} argument {pos + 1} is {got}, the header says {want}")
    assert not bad, "\n".join(bad)
    assert checked >= 100          # (110 at the time of writing: every argument spelled with a conversion, a helper or a cast; handles and nil are not)
