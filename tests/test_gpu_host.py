"""The reference-shaped entry points, end to end over the reference's table formats.

These tests read the way a test of the Go packages would: build forw[]/inv[] tables (md5-hex keys,
JSON rows as in database/noschema_schema.go), call
    ranking.UpdateTopicSensitivePagerank(ctx, 0.75, eps, forw)      (start_crawl.go:175)
    ranking.UpdateTermWeights(ctx, &inv[0], forw, "title")          (:176)
    ranking.UpdateTermWeights(ctx, &inv[1], forw, "body")           (:177)
    retrieval.Retrieve(query, ctx, forw, inv)                       (server.go:47)
through the C++ host mirror (spaghettisearch_amd/host/, module _host) that sits above the C ABI, and
compare the tables / results with the CPU oracle run on the same data.
"""
import hashlib
import json

import numpy as np
import pytest

from spaghettisearch_amd import synth

pytestmark = pytest.mark.gpu


def h(s: str) -> str:
    return hashlib.md5(s.encode()).hexdigest()


@pytest.fixture(scope="module")
def host():
    from spaghettisearch_amd import _lib
    _lib.load()                      # one HIP runtime (torch's) before the module pulls the library in
    from spaghettisearch_amd import _host
    return _host


@pytest.fixture(scope="module")
def corpus():
    """A crawl-shaped corpus (BASELINE config 1: `start_crawl -numPages=1000`): 1000 crawled pages + uncrawled children."""
    rng = np.random.default_rng(2024)
    n_crawled, n_all, n_words = 1000, 1600, 150
    doc = [h(f"http://site/{i}") for i in range(n_all)]
    word = [f"w{i}" for i in range(n_words)]
    children = {}
    for i in range(n_crawled):
        k = int(rng.integers(0, 12))
        children[doc[i]] = [doc[j] for j in sorted(set(rng.integers(0, n_all, size=k).tolist()))]
    cats = {"Arts": {"numPages": 900.0, "wordCount": 1e4}, "Science": {"numPages": 412.0, "wordCount": 5e3},
            "Sports": {"numPages": 77.0, "wordCount": 2e3}}

    uncrawled = sorted({doc.index(c) for v in children.values() for c in v} - set(range(n_crawled)))
    anchor_only = set()

    def inverted(p_doc, seed, anchors):
        """anchors: the table also holds anchor/meta words, which parser.getWordInfo (parser/parser.go:195-207) stores
        as position float32(-100), appended AFTER the real positions and counted in the term frequency; the anchor
        text of a link lands in the TITLE row of the child (indexer.go:261-300), so uncrawled children own title
        postings that consist of -100 entries only."""
        r = np.random.default_rng(seed)
        table = {}
        for wi, w in enumerate(word):
            df = max(1, int(n_crawled * p_doc / (1 + wi * 0.05)))
            ds = r.choice(n_crawled, size=min(df, n_crawled), replace=False)
            row = {}
            for d in ds:
                m = int(r.integers(1, 17))
                c = int(r.integers(1, m + 1))
                pos = sorted(r.integers(0, 500, size=c).astype(float).tolist())
                if anchors and r.random() < 0.3:
                    k = int(r.integers(1, 4))
                    pos += [-100.0] * k                    # not sorted: the reference sorts inside intersect (util.go:187-188)
                    c, m = c + k, m + k
                tf = float(np.float32(c) / np.float32(m))
                row[doc[int(d)]] = [tf] + pos              # [normTF, positions...] (indexer.go:362)
            if anchors:
                for d in r.choice(uncrawled, size=min(len(uncrawled), 1 + wi % 7), replace=False):
                    k = int(r.integers(1, 4))
                    row[doc[int(d)]] = [float(np.float32(k) / np.float32(k + int(r.integers(0, 5))))] + [-100.0] * k
                    anchor_only.add((w, doc[int(d)]))
            table[h(w)] = row
        return table
    return {"doc": doc, "word": word, "children": children, "cats": cats, "anchor_only": anchor_only,
            "title": inverted(0.05, 1, True), "body": inverted(0.4, 2, False), "n_all": n_all}


def make_tables(host, corpus):
    forw = [host.MemDB() for _ in range(6)]
    inv = [host.MemDB() for _ in range(3)]
    for k, v in corpus["children"].items():
        forw[2].set(k, json.dumps(v))
    for k, v in corpus["cats"].items():
        forw[5].set(k, json.dumps(v))
    for k, v in corpus["title"].items():
        inv[0].set(k, json.dumps(v))
    for k, v in corpus["body"].items():
        inv[1].set(k, json.dumps(v))
    return forw, inv


def oracle_graph(corpus):
    names = sorted(set(corpus["children"]) | {c for v in corpus["children"].values() for c in v})
    idx = {k: i for i, k in enumerate(names)}
    ptr = np.zeros(len(names) + 1, dtype=np.uint64)
    for p, c in corpus["children"].items():
        ptr[idx[p] + 1] = len(c)
    ptr = np.cumsum(ptr).astype(np.uint64)
    dst = np.zeros(int(ptr[-1]), dtype=np.uint32)
    for p, c in corpus["children"].items():
        b = int(ptr[idx[p]])
        for j, x in enumerate(c):
            dst[b + j] = idx[x]
    return names, idx, ptr, dst


def oracle_index(table, docs_sorted, terms_sorted):
    didx = {k: i for i, k in enumerate(docs_sorted)}
    ptr, pdoc, ptf = [0], [], []
    for t in terms_sorted:
        row = table.get(t, {})
        ds = sorted(row, key=lambda k: didx[k])
        pdoc += [didx[k] for k in ds]
        ptf += [row[k][0] for k in ds]
        ptr.append(len(pdoc))
    return np.array(ptr, np.uint64), np.array(pdoc, np.uint32), np.array(ptf, np.float32)


def oracle_positions(table, docs_sorted, terms_sorted):
    didx = {k: i for i, k in enumerate(docs_sorted)}
    pos_ptr, pos = [0], []
    for t in terms_sorted:
        row = table.get(t, {})
        for k in sorted(row, key=lambda k: didx[k]):
            pos += row[k][1:]
            pos_ptr.append(len(pos))
    return np.array(pos_ptr, np.uint64), np.array(pos, np.float32)


def test_offline_then_online_like_start_crawl_and_server(host, oracle, corpus):
    forw, inv = make_tables(host, corpus)
    # --- start_crawl.go:175 -------------------------------------------------------------
    host.UpdateTopicSensitivePagerank(0.75, 1e-9, forw)
    names, idx, ptr, dst = oracle_graph(corpus)
    cats = sorted(corpus["cats"])
    ref, ref_iters = oracle.pagerank(len(names), ptr, dst, 0.75, 1e-9, [int(corpus["cats"][c]["numPages"]) for c in cats])
    assert len(forw[3]) == len(names) == corpus["n_all"] or len(forw[3]) == len(names)     # Q1: parents U children
    for v, name in enumerate(names):
        row = json.loads(forw[3].get(name))
        assert sorted(row) == cats
        for k, c in enumerate(cats):
            assert row[c] == pytest.approx(ref[k, v], rel=1e-12)
    # --- start_crawl.go:176-177 ----------------------------------------------------------
    host.UpdateTermWeights(inv[0], forw, "title")
    host.UpdateTermWeights(inv[1], forw, "body")
    terms = sorted(set(corpus["title"]) | set(corpus["body"]))
    mags = {}
    for field, table, t in (("title", corpus["title"], inv[0]), ("body", corpus["body"], inv[1])):
        docs_sorted = sorted(set(names) | {d for row in table.values() for d in row})
        tp, pd, tf = oracle_index(table, docs_sorted, sorted(table))
        w, mag, idf = oracle.tfidf(tp, pd, tf, len(names), len(docs_sorted))         # N = len(forw[3]) (Q7)
        j = 0
        for term in sorted(table):
            row = json.loads(t.get(term))
            for d in sorted(table[term], key=lambda k: docs_sorted.index(k)):
                assert np.float32(row[d][0]) == w[j], (term, d)                         # float32 bit-exact through JSON
                assert row[d][1:] == table[term][d][1:]                                 # positions untouched
                j += 1
        mags[field] = {docs_sorted[i]: mag[i] for i in set(pd.tolist())}
    for d in names:
        row = json.loads(forw[4].get(d)) if d in forw[4].keys() else {}
        for field in ("title", "body"):
            if d in mags[field]:
                assert row[field] == pytest.approx(mags[field][d], rel=1e-12)
    # --- server.go:47 ----------------------------------------------------------------------
    di = host.DeviceIndex()
    di.load(forw, inv)
    queries = ["w3 w17 w40", "W5, w5!  w9", "w149 nosuchword", "zzz", 'w1 "w2 w3" w4', '"w0 w1"', '"w5" w6 "w0"',
               '"w148"', 'w30 "w6"']       # one-term phrases: a doc whose only w148 / w6 posting is anchor text matches through -100
    got = di.RetrieveBatch(queries, 50)
    # oracle on the tables as they are now (weighted), dense ids in sorted key order
    docs_sorted = sorted(set(forw[3].keys()))
    didx = {k: i for i, k in enumerate(docs_sorted)}

    def weighted(t):
        tab = {term: json.loads(t.get(term)) for term in t.keys()}
        return oracle_index(tab, docs_sorted, terms), oracle_positions(tab, docs_sorted, terms)
    (title, tpos), (body, bpos) = weighted(inv[0]), weighted(inv[1])
    mt = np.zeros(len(docs_sorted))
    mb = np.zeros(len(docs_sorted))
    for d in forw[4].keys():
        row = json.loads(forw[4].get(d))
        mt[didx[d]], mb[didx[d]] = row.get("title", 0.0), row.get("body", 0.0)
    tidx = {t: i for i, t in enumerate(terms)}
    for q, res in zip(queries, got):
        import re
        phrases = re.findall(r'"(.*?)"', q)
        rest = q
        for ph in phrases:
            rest = rest.replace('"' + ph + '"', "", 1)
        toks = re.findall(r"[a-z0-9]+", rest.lower())
        ptoks = re.findall(r"[a-z0-9]+", " ".join(phrases).lower())
        qt = np.array([tidx.get(h(t), 0xFFFFFFFF) for t in toks], dtype=np.uint32)
        extra = None
        if ptoks:      # quoted phrase: matched on positions (retrieval/phrase.go), merged at main_retrieve.go:73-78
            extra = oracle.phrase(title, body, tpos, bpos, [tidx[h(t)] for t in ptoks])
        hits, _ = oracle.score_topk(len(docs_sorted), title, body, mt, mb, qt, 50, query_len=len(toks) + len(ptoks), extra=extra)
        assert [r.DocHash for r in res] == [docs_sorted[int(x["doc"])] for x in hits], q
        assert [r.FinalRank for r in res] == hits["final"].tolist()
        assert all(r.PageRank == 0.0 for r in res)                                      # Q9: nil topicProbs
    assert len(got[3]) == 0 and len(got[0]) > 0
    # the -100 sentinel on the Retrieve path: results of the one-term phrases include docs that hold the word as anchor text only
    for qi, w in ((7, "w148"), (8, "w6")):
        only = {d for (ww, d) in corpus["anchor_only"] if ww == w}
        assert only and only & {r.DocHash for r in got[qi]}, (w, len(only))
    # BASELINE config 1: single query, top-10 = the first 10 of the top-50
    top10 = di.RetrieveBatch([queries[0]], 10)[0]
    assert [r.DocHash for r in top10] == [r.DocHash for r in got[0][:10]]
    assert [r.FinalRank for r in top10] == [r.FinalRank for r in got[0][:10]]
    # PageRank blend with explicit topic probabilities (config 5 shape)
    probs = [{"Arts": 0.5, "Science": 0.25, "Sports": 0.25}] * len(queries)
    got_p = di.RetrieveBatch(queries, 50, probs)
    assert di.categories == cats
    r0 = got_p[0][0]
    pr = json.loads(forw[3].get(r0.DocHash))
    assert r0.PageRank == 0.5 * pr["Arts"] + 0.25 * pr["Science"] + 0.25 * pr["Sports"]


def test_md5_matches_hashlib(host):
    for s in ("", "a", "spaghetti", "The quick brown fox jumps over the lazy dog", "x" * 200):
        assert host.md5_hex(s) == hashlib.md5(s.encode()).hexdigest()


def _weighted_tables(host, corpus):
    forw, inv = make_tables(host, corpus)
    host.UpdateTopicSensitivePagerank(0.75, 1e-9, forw)
    host.UpdateTermWeights(inv[0], forw, "title")
    host.UpdateTermWeights(inv[1], forw, "body")
    return forw, inv


def test_snapshot_round_trip(host, corpus, tmp_path):
    """the flattened tables + md5<->dense-id maps go to disk once and a server start loads them without
    touching the JSON tables; answers are identical, field for field."""
    forw, inv = _weighted_tables(host, corpus)
    di = host.DeviceIndex()
    di.load(forw, inv)
    queries = ["w3 w17 w40", 'w1 "w2 w3" w4', "w149 nosuchword", '"w0 w1"']
    probs = [{"Arts": 0.5, "Science": 0.25, "Sports": 0.25}] * len(queries)
    want = di.RetrieveBatch(queries, 50, probs)
    path = str(tmp_path / "corpus.ssnap")
    di.save_snapshot(path)
    assert open(path, "rb").read(8) == b"SSNAP002"
    fresh = host.DeviceIndex()
    fresh.load_snapshot(path)                       # no forw/inv tables involved
    assert fresh.categories == di.categories
    got = fresh.RetrieveBatch(queries, 50, probs)
    for a, b in zip(want, got):
        assert [(r.DocHash, r.FinalRank, r.PageRank, r.TitleRank, r.BodyRank) for r in a] == \
               [(r.DocHash, r.FinalRank, r.PageRank, r.TitleRank, r.BodyRank) for r in b]
    # a damaged file is refused, not half-loaded
    blob = open(path, "rb").read()
    bad = tmp_path / "bad.ssnap"
    bad.write_bytes(b"NOTASNAP" + blob[8:])
    with pytest.raises(RuntimeError):
        host.DeviceIndex().load_snapshot(str(bad))
    bad.write_bytes(blob[:len(blob) // 2])
    with pytest.raises(RuntimeError):
        host.DeviceIndex().load_snapshot(str(bad))


def test_concurrent_requests_are_batched(host, corpus):
    """cmd/server/server.go:47 serves one goroutine per request; the batching front-end answers concurrent callers with one
    library call and hands every caller its own result."""
    import threading
    forw, inv = _weighted_tables(host, corpus)
    di = host.DeviceIndex()
    di.load(forw, inv)
    queries = [f"w{i % 150} w{(7 * i + 3) % 150} w{(13 * i + 5) % 150}" for i in range(64)]
    want = di.RetrieveBatch(queries, 50)
    batcher = host.RetrieveBatcher(di, 50, 20000, 1024)      # generous window: the threads below all make it into few batches
    got = [None] * len(queries)

    def worker(i):
        got[i] = batcher.Retrieve(queries[i])

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(queries))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, b in zip(want, got):
        assert [(r.DocHash, r.FinalRank) for r in a] == [(r.DocHash, r.FinalRank) for r in b]
    assert batcher.batches < len(queries) and batcher.largest_batch > 1
    one = batcher.Retrieve(queries[0])                       # a lone caller is served after the window
    assert [(r.DocHash, r.FinalRank) for r in one] == [(r.DocHash, r.FinalRank) for r in want[0]]
