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
    # the same call with every category's ranks from two vectors (TopicSensitive::two_vectors, library option pr.affine)
    forw2, _inv2 = make_tables(host, corpus)
    host.UpdateTopicSensitivePagerank(0.75, 1e-9, forw2, two_vectors=True)
    for v, name in enumerate(names):
        row = json.loads(forw2[3].get(name))
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


def test_batcher_keeps_batches_in_flight(host, corpus):
    """Under a steady stream of callers the batcher enqueues the next batch (ss_score_topk_submit) before it hands the previous one back
    (ss_score_topk_collect): 24 threads x 12 requests with a short window make many overlapping batches, quoted phrases included;
    every caller still gets exactly its own result."""
    import threading
    forw, inv = _weighted_tables(host, corpus)
    di = host.DeviceIndex()
    di.load(forw, inv)
    queries = [(f'w{i % 150} "w{(5 * i + 1) % 150} w{(5 * i + 2) % 150}" w{(11 * i + 7) % 150}' if i % 3 == 0
                else f"w{i % 150} w{(7 * i + 3) % 150} w{(13 * i + 5) % 150}") for i in range(24 * 12)]
    want = di.RetrieveBatch(queries, 50)
    batcher = host.RetrieveBatcher(di, 50, 200, 64)
    got = [None] * len(queries)

    def worker(t):
        for j in range(12):
            i = t * 12 + j
            got[i] = batcher.Retrieve(queries[i])

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, b in zip(want, got):
        assert [(r.DocHash, r.FinalRank, r.TitleRank, r.BodyRank) for r in a] == [(r.DocHash, r.FinalRank, r.TitleRank, r.BodyRank) for r in b]
    assert 12 <= batcher.batches < len(queries)


def _retrieve_key(res):
    """sort.Slice on FinalRank is unstable in the reference (main_retrieve.go:96) and dense ids differ between a patched
    and a freshly loaded index: compare up to the order of equal FinalRanks."""
    return sorted((-r.FinalRank, r.DocHash, r.TitleRank, r.BodyRank, r.PageRank) for r in res)


def test_recrawl_one_page_apply_delta(host, oracle, corpus):
    """SURVEY.md §8f-4 end to end: a page changed since the last crawl (indexer.go:420-641 checkAndUpdate, then the re-index
    :23-348).  The tables get the reference's writes, the resident device tables ONE delta each, the resident link graph
    its changed row — and Retrieve answers like an index loaded from the updated tables, and like the oracle on them."""
    forw, inv = _weighted_tables(host, corpus)
    doc, word = corpus["doc"], corpus["word"]
    di = host.DeviceIndex()
    di.load(forw, inv)
    pr = host.ResidentPagerank()
    pr.Build(forw)
    page = doc[17]
    # what the indexer holds of the page as crawled before: its title / body words and the anchor words of its links
    before = {"docHash": page,
              "title": {t: row[page] for t, row in corpus["title"].items() if page in row},
              "body": {t: row[page] for t, row in corpus["body"].items() if page in row},
              "children": corpus["children"][page], "anchors": {}}
    kids_old = corpus["children"][page]
    assert kids_old and before["body"]
    # anchor words this page put into an old child's title row: pick postings that exist (the reference panics on a missing row)
    for c in kids_old[:2]:
        ws = [t for t, row in corpus["title"].items() if c in row][:2]
        if ws:
            before["anchors"][c] = {t: json.loads(inv[0].get(t))[c] for t in ws}
    # the page as crawled now: other words (one brand new), other links (one to a page never seen), new anchor texts
    new_word, new_child = h("brandnewword"), h("http://site/never-seen-before")
    kept_child = kids_old[0]
    after = {"docHash": page,
             "title": {h(word[3]): [1.0, 0.0], h(word[17]): [0.5, 1.0, -100.0]},
             "body": {h(word[3]): [0.25, 4.0, 9.0], h(word[40]): [1.0, 0.0, 1.0, 2.0, 7.0], new_word: [0.5, 3.0, 5.0]},
             "children": [kept_child, doc[5], new_child],
             "anchors": {kept_child: {h(word[9]): [1.0, -100.0]},
                         new_child: {new_word: [1.0, -100.0, -100.0], h(word[3]): [0.5, -100.0]}}}
    di.ApplyDelta(forw, inv, before, after)
    # --- the tables hold what the reference's writes leave -----------------------------------------------------------------
    for t, lp in before["title"].items():
        row = json.loads(inv[0].get(t)) if t in inv[0].keys() else {}
        assert (page in row) == (t in after["title"])
    for t in before["body"]:
        row = json.loads(inv[1].get(t)) if t in inv[1].keys() else {}
        assert (page in row) == (t in after["body"])
    for c, ws in before["anchors"].items():
        for t in ws:
            row = json.loads(inv[0].get(t)) if t in inv[0].keys() else {}
            assert (c in row) == (t in after["anchors"].get(c, {}))
    assert json.loads(inv[1].get(new_word)) == {page: [0.5, 3.0, 5.0]}
    assert json.loads(inv[0].get(new_word)) == {new_child: [1.0, -100.0, -100.0]}
    assert json.loads(forw[2].get(page)) == after["children"]
    # --- magnitudes of the touched docs: forw[4] rows rewritten from the device's O(delta) update ---------------------------
    for d in [page, new_child, kept_child] + list(before["anchors"]):
        row = json.loads(forw[4].get(d))
        for field, t in (("title", inv[0]), ("body", inv[1])):
            sq = 0.0
            for term in t.keys():
                lp = json.loads(t.get(term)).get(d)
                if lp is not None:
                    sq += float(np.float32(lp[0]) * np.float32(lp[0]))
            assert row[field] == pytest.approx(np.sqrt(sq), rel=1e-12), (d, field)
    # --- Retrieve: patched index == index loaded from the updated tables == oracle ------------------------------------------
    queries = ["w3 w17 w40", "brandnewword w9", '"w3"', 'w40 "w0 w1"', "w5 w6 w7", '"brandnewword"']
    got = di.RetrieveBatch(queries, 1000)
    fresh = host.DeviceIndex()
    fresh.load(forw, inv)
    want = fresh.RetrieveBatch(queries, 1000)
    for q, a, b in zip(queries, got, want):
        assert len(a) == len(b) and len(a) > 0, q
        assert _retrieve_key(a) == _retrieve_key(b), q
    assert page in {r.DocHash for r in got[1]} and new_child in {r.DocHash for r in got[5]}
    # oracle on the updated tables (dense ids in sorted key order)
    docs_sorted = sorted(set(forw[3].keys()) | {d for t in (inv[0], inv[1]) for term in t.keys() for d in json.loads(t.get(term))})
    terms = sorted(set(inv[0].keys()) | set(inv[1].keys()))
    didx = {k: i for i, k in enumerate(docs_sorted)}
    tidx = {t: i for i, t in enumerate(terms)}

    def tab(t):
        rows = {term: json.loads(t.get(term)) for term in t.keys()}
        return oracle_index(rows, docs_sorted, terms)
    title, body = tab(inv[0]), tab(inv[1])
    mt, mb = np.zeros(len(docs_sorted)), np.zeros(len(docs_sorted))
    for d in forw[4].keys():
        row = json.loads(forw[4].get(d))
        mt[didx[d]], mb[didx[d]] = row.get("title", 0.0), row.get("body", 0.0)
    for q, res in ((queries[0], got[0]), (queries[1], got[1]), (queries[4], got[4])):
        toks = q.split()
        qt = np.array([tidx.get(h(t), 0xFFFFFFFF) for t in toks], dtype=np.uint32)
        hits, _ = oracle.score_topk(len(docs_sorted), title, body, mt, mb, qt, 1000, query_len=len(toks))
        assert sorted((-float(x["final"]), docs_sorted[int(x["doc"])]) for x in hits) == [(a, b) for a, b, *_ in _retrieve_key(res)], q
    # a snapshot written after the delta carries the patched tables
    import tempfile, os
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "after.ssnap")
        di.save_snapshot(path)
        snap = host.DeviceIndex()
        snap.load_snapshot(path)
        for a, b in zip(got, snap.RetrieveBatch(queries, 1000)):
            assert _retrieve_key(a) == _retrieve_key(b)
    # --- the link graph: one changed row patched on the device, PageRank re-run, forw[3] rewritten -------------------------
    pr.ApplyDelta(forw, [page])
    pr.Run(0.75, 1e-9, forw)
    children = dict(corpus["children"])
    children[page] = after["children"]
    names, idx, ptr, dst = oracle_graph({"children": children})
    cats = sorted(corpus["cats"])
    ref, _ = oracle.pagerank(len(names), ptr, dst, 0.75, 1e-9, [int(corpus["cats"][c]["numPages"]) for c in cats])
    assert sorted(forw[3].keys()) == names and new_child in names
    for v, name in enumerate(names):
        row = json.loads(forw[3].get(name))
        for k, c in enumerate(cats):
            assert row[c] == pytest.approx(ref[k, v], rel=1e-11)
    # the scorer's PageRank table follows
    di.ReloadPrior(forw)
    probs = [{"Arts": 0.5, "Science": 0.25, "Sports": 0.25}]
    r = [x for x in di.RetrieveBatch(['"brandnewword"'], 50, probs)[0] if x.DocHash == new_child][0]
    row = json.loads(forw[3].get(new_child))
    assert r.PageRank == 0.5 * row["Arts"] + 0.25 * row["Science"] + 0.25 * row["Sports"]


def test_resident_pagerank_drops_a_page_nobody_links_to_any_more(host, oracle, corpus):
    """the reference rebuilds its node set = parents U children from forw[2] on every run (pagerank.go:17-44) and
    the node COUNT enters every rank (pagerank.go:111).  A child that only the re-crawled page linked to, and that is no
    parent itself, must leave the resident graph when the link goes — not stay behind as an isolated node."""
    forw, inv = make_tables(host, corpus)
    doc = corpus["doc"]
    children = dict(corpus["children"])
    indeg = {}
    for p, cs in children.items():
        for c in cs:
            indeg[c] = indeg.get(c, 0) + 1
    page, lonely = next((p, c) for p, cs in sorted(children.items()) for c in cs if indeg[c] == 1 and c not in children)
    pr = host.ResidentPagerank()
    pr.Build(forw)
    assert lonely in pr.name
    cats = sorted(corpus["cats"])
    n_topic = [int(corpus["cats"][c]["numPages"]) for c in cats]

    def check_against_oracle(ch):
        names, idx, ptr, dst = oracle_graph({"children": ch})
        ref, _ = oracle.pagerank(len(names), ptr, dst, 0.75, 1e-9, n_topic)
        assert sorted(pr.name) == names
        for v, name in enumerate(names):
            row = json.loads(forw[3].get(name))
            for k, c in enumerate(cats):
                assert row[c] == pytest.approx(ref[k, v], rel=1e-11)
    # 1. the page drops its link to `lonely`: the node set shrinks, the resident graph is re-flattened
    children[page] = [c for c in children[page] if c != lonely]
    forw[2].set(page, json.dumps(children[page]))
    pr.ApplyDelta(forw, [page])
    assert pr.rebuilds == 1 and lonely not in pr.name
    pr.Run(0.75, 1e-9, forw)
    check_against_oracle(children)
    # 2. a delta that orphans nobody is patched on the device (no rebuild), also when it brings a new page
    newcomer = h("http://site/newcomer")
    children[page] = children[page] + [newcomer, doc[3]]
    forw[2].set(page, json.dumps(children[page]))
    pr.ApplyDelta(forw, [page])
    assert pr.rebuilds == 1 and newcomer in pr.name
    pr.Run(0.75, 1e-9, forw)
    check_against_oracle(children)
    # 3. ... and the newcomer goes again with its only link
    children[page] = [c for c in children[page] if c != newcomer]
    forw[2].set(page, json.dumps(children[page]))
    pr.ApplyDelta(forw, [page])
    assert pr.rebuilds == 2 and newcomer not in pr.name
    pr.Run(0.75, 1e-9, forw)
    check_against_oracle(children)


def test_opt_in_topic_sensitive_wiring(host, oracle, corpus):
    """SURVEY.md §8f-3 above the C ABI: OFF by default (reference-identical tables and answers), ON = teleport sets from the
    stored ODP keyword vectors (forw[5], inv[2]) for UpdateTopicSensitivePagerank and live computeTopicProbs for Retrieve."""
    forw, inv = make_tables(host, corpus)
    cats = sorted(corpus["cats"])
    rng = np.random.default_rng(8)
    # ODP keyword vectors (crawler/ODP-scraper.go:97-139): word -> {category: frequency}
    kw = {}
    for wi in rng.choice(len(corpus["word"]), size=60, replace=False):
        cs = rng.choice(cats, size=int(rng.integers(1, 3)), replace=False)
        kw[h(corpus["word"][int(wi)])] = {str(c): int(rng.integers(1, 40)) for c in cs}
    for k, v in kw.items():
        inv[2].set(k, json.dumps(v))
    # default: the reference's uniform teleport, whatever inv[2] holds
    host.UpdateTopicSensitivePagerank(0.75, 1e-9, forw)
    names, idx, ptr, dst = oracle_graph(corpus)
    n_topic = [int(corpus["cats"][c]["numPages"]) for c in cats]
    ref, _ = oracle.pagerank(len(names), ptr, dst, 0.75, 1e-9, n_topic)
    for v in (0, 7, len(names) - 1):
        row = json.loads(forw[3].get(names[v]))
        assert [row[c] for c in cats] == pytest.approx(ref[:, v].tolist(), rel=1e-12)
    default_rows = {k: forw[3].get(k) for k in forw[3].keys()}
    host.UpdateTopicSensitivePagerank(0.75, 1e-9, forw, False, inv)
    assert {k: forw[3].get(k) for k in forw[3].keys()} == default_rows                    # explicit off == default, bit for bit
    # opt-in: sets by the stated rule, restated here in Python
    wc = {c: corpus["cats"][c]["wordCount"] for c in cats}
    mass = {}
    for w in sorted(kw):
        share = [kw[w].get(c, 0) / wc[c] for c in cats]
        for table in (corpus["title"], corpus["body"]):
            for d in table.get(w, {}):
                m = mass.setdefault(d, [0.0] * len(cats))
                for i in range(len(cats)):
                    m[i] += share[i]
    want_sets = {c: [] for c in cats}
    for d in sorted(mass):
        best = int(np.argmax(mass[d]))                                                     # first maximum, like the mirror
        if mass[d][best] > 0:
            want_sets[cats[best]].append(d)
    got_sets = host.TopicTeleportSets(forw, inv)
    assert got_sets == want_sets and all(len(v) > 0 for v in want_sets.values())
    host.UpdateTopicSensitivePagerank(0.75, 1e-9, forw, True, inv)
    for k, c in enumerate(cats):
        members = [idx[d] for d in want_sets[c] if d in idx]
        ref_ts, _ = oracle.pagerank_topic_ts(len(names), ptr, dst, 0.75, 1e-9, n_topic[k], np.array(members, dtype=np.uint32))
        got = np.array([json.loads(forw[3].get(nm))[c] for nm in names])
        np.testing.assert_allclose(got, ref_ts, rtol=1e-11, atol=1e-300)
        assert got[members].mean() > 1.5 * ref[k][members].mean()                         # the set's pages gained rank
    # Retrieve: off by default; live probabilities = fixed computeTopicProbs over the query's non-phrase words
    host.UpdateTermWeights(inv[0], forw, "title")
    host.UpdateTermWeights(inv[1], forw, "body")
    di = host.DeviceIndex()
    di.load(forw, inv)
    di.LoadTopics(forw, inv)
    kws = [w for w in corpus["word"] if h(w) in kw]
    plain = [w for w in corpus["word"] if h(w) not in kw]
    queries = [f"{kws[0]} {kws[1]}", f"{kws[2]} {plain[0]}", f"{plain[1]} {plain[2]}", f'{kws[3]} "{kws[4]}"']
    off = di.RetrieveBatch(queries, 50)
    assert all(r.PageRank == 0.0 for res in off for r in res)                             # Q9: nil topicProbs
    on = di.RetrieveBatch(queries, 50, None, True)
    wcl = [wc[c] for c in cats]
    for q, res_on, res_off in zip(queries, on, off):
        toks = [t for t in q.replace('"' + kws[4] + '"', "").split()]
        maps = [{cats.index(c): f for c, f in kw[h(t)].items()} for t in toks if h(t) in kw]
        probs = oracle.topic_probs(wcl, maps, mode=1) if maps else np.zeros(len(cats))
        assert list(di.liveTopicProbs([h(t) for t in toks]).values()) == probs.tolist()
        explicit = di.RetrieveBatch([q], 50, [dict(zip(cats, probs.tolist()))])[0]
        assert [(r.DocHash, r.FinalRank, r.PageRank) for r in res_on] == [(r.DocHash, r.FinalRank, r.PageRank) for r in explicit]
        for r in res_on[:5]:
            row = json.loads(forw[3].get(r.DocHash))
            assert r.PageRank == pytest.approx(sum(p * row[c] for p, c in zip(probs.tolist(), cats)), rel=1e-12)
    assert any(r.PageRank > 0 for r in on[0]) and all(r.PageRank == 0.0 for r in on[2])


def test_one_bad_request_does_not_fail_its_batch(host, corpus):
    """A quoted phrase beyond SS_MAX_PHRASE_TERMS makes the library refuse the whole batch: the batching front-end then answers
    the callers one by one, and only the offending caller sees the error."""
    import threading
    forw, inv = _weighted_tables(host, corpus)
    di = host.DeviceIndex()
    di.load(forw, inv)
    good = [f"w{i} w{i + 1}" for i in range(8)]
    bad = '"' + " ".join(f"w{i}" for i in range(20)) + '"'
    want = di.RetrieveBatch(good, 50)
    with pytest.raises(RuntimeError):
        di.RetrieveBatch(good + [bad], 50)
    batcher = host.RetrieveBatcher(di, 50, 50000, 1024)
    queries = good + [bad]
    got, errs = [None] * len(queries), [None] * len(queries)

    def worker(i):
        try:
            got[i] = batcher.Retrieve(queries[i])
        except Exception as e:          # noqa: BLE001 - the error of the offending request
            errs[i] = e

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(queries))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errs[-1] is not None and all(e is None for e in errs[:-1])
    for a, b in zip(want, got[:-1]):
        assert [(r.DocHash, r.FinalRank) for r in a] == [(r.DocHash, r.FinalRank) for r in b]
