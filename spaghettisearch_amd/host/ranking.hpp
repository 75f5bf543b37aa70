// ranking.hpp — host-side mirror of the reference's package `ranking` above the C ABI.
//
// Same names, argument meaning and error behaviour as
//   ranking.UpdateTopicSensitivePagerank(ctx, dampingFactor, convergenceCriterion, forward)  pagerank.go:14
//   ranking.UpdateTermWeights(ctx, inv, forw, info)                                          term_weighting.go:10
// (the reference panics on every error: these throw std::runtime_error).  The bodies are the
// table <-> flat-array bridge of SURVEY.md §8f-2: JSON rows keyed by md5-hex strings are flattened to
// dense-id CSR arrays, ONE library call does the arithmetic on the GPU, results are written back in the
// reference's table formats.  Go is not available in this image, so this C++ layer is the compiled
// counterpart of go/ranking/ranking.go.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/spaghetti_rank.h"
#include "database.hpp"
#include "json_mini.hpp"

namespace spaghetti {

// process-wide library context on GPU 0 (one process per GPU)
inline ss_ctx* default_ctx() {
    static ss_ctx* ctx = [] {
        ss_ctx* c = nullptr;
        const int32_t rc = ss_init(0, &c);
        if (rc != SS_OK) throw std::runtime_error(std::string("ss_init: ") + ss_last_error(nullptr));
        return c;
    }();
    return ctx;
}
inline void check(int32_t rc, const char* what) {
    if (rc != SS_OK) throw std::runtime_error(std::string(what) + ": " + ss_last_error(default_ctx()));   // reference: panic(err)
}

// md5-hex keys -> dense ids in sorted key order (deterministic, unlike Go map iteration)
struct DenseIds {
    std::vector<std::string> name;
    std::unordered_map<std::string, uint32_t> id;
    template <typename It>
    void build(It first, It last) {
        name.assign(first, last);
        std::sort(name.begin(), name.end());
        name.erase(std::unique(name.begin(), name.end()), name.end());
        id.reserve(name.size() * 2);
        for (size_t i = 0; i < name.size(); i++) id[name[i]] = (uint32_t)i;
    }
};

}  // namespace spaghetti

namespace ranking {

// OPT-IN, beyond what the reference executes.  The reference advertises topic-sensitive PageRank
// (README.md:9) but its topics differ only by the start value 1/numPages (pagerank.go:54-63,104); the topic data it does
// store are the ODP keyword vectors: forw[5][category] = {numPages, wordCount}, inv[2][wordHash] = map[category]frequency
// (crawler/ODP-scraper.go:97-139).  With `teleport_sets` a crawled page joins the teleport set of the category whose
// keywords it holds most of, by the estimate computeTopicProbs uses for a query (tf / wordCount, main_retrieve.go:143-145):
//     mass(page, c) = sum over the words w of the page (inv[0] or inv[1] row of w holds the page) of inv[2][w][c] / wordCount(c)
// topic(page) = argmax_c mass (ties: the first category in key order); a page without keyword hits joins no set, and a
// category without pages keeps the reference's uniform teleport (ss_pr_set_teleport).  Default: off = reference behaviour.
struct TopicSensitive {
    bool teleport_sets = false;
    std::vector<db::DB*>* inv = nullptr;      // needed with teleport_sets: inv[0], inv[1], inv[2]
    // opt-in: every category's ranks from TWO vectors (library option "pr.affine": the reference's categories differ only in their
    // start value 1/numPages, and its recurrence maps a ratio of affine forms in it onto itself).  Same ranks to ~1e-15, not the
    // reference's operation order; the cost no longer grows with the number of categories.  Not with teleport_sets.
    bool two_vectors = false;
};

// category -> the doc hashes of its teleport set (sorted), by the rule above
inline std::map<std::string, std::vector<std::string>> TopicTeleportSets(db::Context& ctx, std::vector<db::DB*>& forward,
                                                                         std::vector<db::DB*>& inv) {
    std::vector<std::string> cat;
    std::vector<double> word_count;
    for (auto& kv : forward[5]->Iterate(ctx)) {
        auto md = jsonmini::parse_map_f64(kv.second);
        cat.push_back(kv.first);
        word_count.push_back(md.count("wordCount") ? md["wordCount"] : 0.0);
    }
    const size_t K = cat.size();
    std::map<std::string, std::vector<double>> mass;                      // doc -> [K]
    for (auto& kw : inv[2]->Iterate(ctx)) {
        const std::map<std::string, double> freq = jsonmini::parse_map_f64(kw.second);
        std::vector<double> share(K, 0.0);
        bool any = false;
        for (size_t c = 0; c < K; c++) {
            auto it = freq.find(cat[c]);
            if (it != freq.end() && word_count[c] > 0.0) { share[c] = it->second / word_count[c]; any = true; }
        }
        if (!any) continue;
        for (int table = 0; table < 2; table++) {
            if (!inv[table]->Has(ctx, kw.first)) continue;
            for (auto& post : jsonmini::parse_map_f32list(inv[table]->Get(ctx, kw.first))) {
                auto& m = mass[post.first];
                if (m.empty()) m.assign(K, 0.0);
                for (size_t c = 0; c < K; c++) m[c] += share[c];
            }
        }
    }
    std::map<std::string, std::vector<std::string>> sets;
    for (auto& c : cat) sets[c];
    for (auto& dm : mass) {
        size_t best = 0;
        for (size_t c = 1; c < K; c++)
            if (dm.second[c] > dm.second[best]) best = c;
        if (dm.second[best] > 0.0) sets[cat[best]].push_back(dm.first);   // std::map iteration: doc hashes arrive sorted
    }
    return sets;
}

// The link graph kept on the device between crawls: Build flattens forw[2] once, ApplyDelta patches the
// resident adjacency with the re-crawled parents' new rows (ss_graph_apply_delta: no re-flatten, no re-upload), Run is the
// compute + write-back half of UpdateTopicSensitivePagerank.  New pages get ids at the end, so ids are stable across deltas.
// The reference rebuilds its node set = parents U children from forw[2] on every run (pagerank.go:17-44), and the node COUNT
// enters every rank (totalValue = ... + teleportProbs * len(currentRank), :111): a page that a delta leaves neither a parent
// nor anybody's child must leave the node set.  The host therefore keeps every parent's child list and every node's
// in-degree; a delta that orphans a node falls back to Build (ids are compacted, `rebuilds` counts it).
class ResidentPagerank {
public:
    std::vector<std::string> name;
    std::unordered_map<std::string, uint32_t> id;
    ss_graph* g = nullptr;
    std::vector<std::vector<uint32_t>> kids_of;      // [node] children as uploaded (empty for frontier pages)
    std::vector<uint8_t> is_parent;                  // [node] has a forw[2] row
    std::vector<uint32_t> indeg;                     // [node] references from the child lists
    int rebuilds = 0;                                // ApplyDelta calls that had to re-flatten (a node was orphaned)

    ResidentPagerank() = default;
    ResidentPagerank(const ResidentPagerank&) = delete;
    ResidentPagerank& operator=(const ResidentPagerank&) = delete;
    ~ResidentPagerank() { if (g) ss_graph_destroy(g); }

    // pagerank.go:17-44 — node set = parents U children (frontier pages are nodes without children)
    void Build(db::Context& ctx, std::vector<db::DB*>& forward) {
        using namespace spaghetti;
        if (g) { ss_graph_destroy(g); g = nullptr; }
        const std::vector<db::KV> nodes = forward[2]->Iterate(ctx);
        std::vector<std::vector<std::string>> children(nodes.size());
        std::vector<std::string> all;
        for (size_t i = 0; i < nodes.size(); i++) {
            children[i] = jsonmini::parse_string_list(nodes[i].second);
            all.push_back(nodes[i].first);
            for (auto& c : children[i]) all.push_back(c);
        }
        DenseIds ids;
        ids.build(all.begin(), all.end());
        name = std::move(ids.name);
        id = std::move(ids.id);
        const size_t n = name.size();
        if (n == 0) return;
        std::vector<uint64_t> out_ptr(n + 1, 0);
        for (size_t i = 0; i < nodes.size(); i++) out_ptr[id[nodes[i].first] + 1] = children[i].size();
        for (size_t v = 0; v < n; v++) out_ptr[v + 1] += out_ptr[v];
        std::vector<uint32_t> out_dst(out_ptr[n]);
        kids_of.assign(n, {});
        is_parent.assign(n, 0);
        indeg.assign(n, 0);
        for (size_t i = 0; i < nodes.size(); i++) {
            const uint32_t p = id[nodes[i].first];
            uint64_t base = out_ptr[p];
            is_parent[p] = 1;
            kids_of[p].reserve(children[i].size());
            for (auto& c : children[i]) {
                const uint32_t v = id[c];
                out_dst[base++] = v;
                kids_of[p].push_back(v);
                indeg[v]++;
            }
        }
        check(ss_graph_create(default_ctx(), n, out_dst.size(), out_ptr.data(), out_dst.data(), 0, 1, &g), "ss_graph_create");
    }

    // The forw[2] rows of `parents` were rewritten (indexer.go:301-304 after a re-crawl): patch the resident graph.
    void ApplyDelta(db::Context& ctx, std::vector<db::DB*>& forward, const std::vector<std::string>& parents) {
        using namespace spaghetti;
        if (!g) { Build(ctx, forward); return; }
        auto node = [&](const std::string& h) {
            auto it = id.find(h);
            if (it != id.end()) return it->second;
            const uint32_t v = (uint32_t)name.size();
            name.push_back(h);
            id[h] = v;
            return v;
        };
        std::vector<std::string> uniq(parents);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        const size_t n_before = name.size();
        std::vector<uint32_t> changed, kids;
        std::vector<uint64_t> ptr{0};
        std::vector<uint8_t> has_row;
        for (auto& p : uniq) {
            changed.push_back(node(p));
            has_row.push_back(forward[2]->Has(ctx, p) ? 1 : 0);
            if (has_row.back())
                for (auto& c : jsonmini::parse_string_list(forward[2]->Get(ctx, p))) kids.push_back(node(c));
            ptr.push_back(kids.size());
        }
        // would the delta leave a node that is neither a parent nor a child?  (counted on copies: nothing changes on failure)
        std::vector<uint32_t> deg(indeg);
        deg.resize(name.size(), 0);
        std::vector<uint8_t> par(is_parent);
        par.resize(name.size(), 0);
        for (size_t i = 0; i < changed.size(); i++) {
            if (changed[i] < n_before)
                for (uint32_t c : kids_of[changed[i]]) deg[c]--;
            for (uint64_t j = ptr[i]; j < ptr[i + 1]; j++) deg[kids[j]]++;
            par[changed[i]] = has_row[i];
        }
        bool orphan = false;
        for (size_t i = 0; i < changed.size() && !orphan; i++) {
            orphan = !par[changed[i]] && deg[changed[i]] == 0;
            if (changed[i] < n_before)
                for (uint32_t c : kids_of[changed[i]]) orphan = orphan || (!par[c] && deg[c] == 0);
        }
        if (orphan) {                              // the node set shrinks: re-flatten (pagerank.go:17-44 does so on every run)
            rebuilds++;
            Build(ctx, forward);
            return;
        }
        const int32_t rc = ss_graph_apply_delta(g, name.size(), changed.size(), changed.data(), ptr.data(), kids.data());
        if (rc != SS_OK) {                         // the graph is unchanged: so are the ids
            for (size_t v = n_before; v < name.size(); v++) id.erase(name[v]);
            name.resize(n_before);
        }
        check(rc, "ss_graph_apply_delta");
        indeg = std::move(deg);
        is_parent = std::move(par);
        kids_of.resize(name.size());
        for (size_t i = 0; i < changed.size(); i++) kids_of[changed[i]].assign(kids.begin() + ptr[i], kids.begin() + ptr[i + 1]);
    }

    // pagerank.go:46-82 — one power iteration per category (all of them in one K-wide run), forw[3][doc] = map[category]rank
    void Run(db::Context& ctx, double dampingFactor, double convergenceCriterion, std::vector<db::DB*>& forward,
             const TopicSensitive& ts = TopicSensitive()) {
        using namespace spaghetti;
        const size_t n = name.size();
        if (n == 0) return;
        const std::vector<db::KV> cats = forward[5]->Iterate(ctx);
        std::vector<std::string> cat_name;
        std::vector<int32_t> n_topic;
        for (auto& kv : cats) {
            auto val = jsonmini::parse_map_f64(kv.second);
            std::fprintf(stderr, "number of webnodes in %s is %d\n", kv.first.c_str(), (int)val["numPages"]);
            cat_name.push_back(kv.first);
            n_topic.push_back((int32_t)(int)val["numPages"]);
        }
        const int K = (int)cat_name.size();
        std::vector<double> rank((size_t)K * n);
        if (K > 0 && !ts.teleport_sets) {
            std::vector<int32_t> iters(K);
            if (ts.two_vectors) check(ss_set_option(default_ctx(), "pr.affine", 1), "ss_set_option");
            const int32_t rc = ss_pagerank_run(g, dampingFactor, convergenceCriterion, 0, K, n_topic.data(), rank.data(), iters.data());
            if (ts.two_vectors) (void)ss_set_option(default_ctx(), "pr.affine", SS_OPTION_DEFAULT);
            check(rc, "ss_pagerank_run");
        } else if (K > 0) {
            if (!ts.inv) throw std::runtime_error("UpdateTopicSensitivePagerank: teleport_sets needs the inverted tables");
            const auto sets = TopicTeleportSets(ctx, forward, *ts.inv);
            std::vector<uint64_t> set_ptr{0};
            std::vector<uint32_t> set_nodes;
            for (auto& c : cat_name) {
                for (auto& d : sets.at(c)) {
                    auto it = id.find(d);
                    if (it != id.end()) set_nodes.push_back(it->second);   // a page outside the link graph has no rank to receive
                }
                set_ptr.push_back(set_nodes.size());
            }
            ss_pr* pr = nullptr;
            check(ss_pr_create(g, dampingFactor, convergenceCriterion, 0, K, n_topic.data(), &pr), "ss_pr_create");
            int32_t rc = ss_pr_set_teleport(pr, set_ptr.data(), set_nodes.data());
            if (rc == SS_OK) rc = ss_pr_begin(pr);
            int32_t n_active = K, sweeps = 0;
            std::vector<int32_t> iters(K);
            while (rc == SS_OK && n_active > 0) {
                rc = ss_pr_step(pr, 8);                          // the stop rule runs on the device: finished topics stand still
                if (rc == SS_OK) rc = ss_pr_status(pr, iters.data(), &n_active, &sweeps, nullptr, nullptr);
            }
            if (rc == SS_OK) rc = ss_pr_read(pr, rank.data());
            ss_pr_destroy(pr);
            check(rc, "topic-sensitive PageRank");
        }
        auto bw = forward[3]->BatchWrite_init(ctx);
        for (size_t v = 0; v < n; v++) {
            std::map<std::string, double> PR;
            for (int k = 0; k < K; k++) PR[cat_name[k]] = rank[(size_t)k * n + v];
            bw->BatchSet(ctx, name[v], jsonmini::dump(PR));
        }
        bw->Flush(ctx);
    }
};

inline void UpdateTopicSensitivePagerank(db::Context& ctx, double dampingFactor, double convergenceCriterion,
                                         std::vector<db::DB*>& forward, const TopicSensitive& ts = TopicSensitive()) {
    std::fprintf(stderr, "Ranking with damping factor='%f', convergence_criteria='%f'\n", dampingFactor, convergenceCriterion);
    ResidentPagerank pr;
    pr.Build(ctx, forward);
    pr.Run(ctx, dampingFactor, convergenceCriterion, forward, ts);
}

// term_weighting.go:59-123
inline void saveMagnitude(db::Context& ctx, std::map<std::string, double>& pageMagnitude, db::DB* forw, const std::string& info) {
    const std::vector<db::KV> comp = forw->Iterate(ctx);
    auto bw = forw->BatchWrite_init(ctx);
    for (auto& kv : comp) {
        auto v = jsonmini::parse_map_f64(kv.second);
        auto it = pageMagnitude.find(kv.first);
        v[info] = it == pageMagnitude.end() ? 0.0 : it->second;    // math.Sqrt(pageMagnitude[key]) of a missing key = 0 (:97)
        if (it != pageMagnitude.end()) pageMagnitude.erase(it);   // :98
        bw->BatchSet(ctx, kv.first, jsonmini::dump(v));
    }
    for (auto& kv : pageMagnitude) {                               // :102-106 / base case :67-77
        std::map<std::string, double> v{{info, kv.second}};
        bw->BatchSet(ctx, kv.first, jsonmini::dump(v));
    }
    bw->Flush(ctx);
}

inline void UpdateTermWeights(db::Context& ctx, db::DB** inv, std::vector<db::DB*>& forw, const std::string& info) {
    using namespace spaghetti;
    // term_weighting.go:12-17 — N = number of PageRank nodes
    const std::vector<db::KV> ranks = forw[3]->Iterate(ctx);
    const uint64_t totalDocs = ranks.size();
    const std::vector<db::KV> comp = (*inv)->Iterate(ctx);
    std::vector<std::map<std::string, std::vector<float>>> rows(comp.size());
    std::vector<std::string> all;
    for (auto& kv : ranks) all.push_back(kv.first);
    for (size_t i = 0; i < comp.size(); i++) {
        rows[i] = jsonmini::parse_map_f32list(comp[i].second);
        for (auto& kv : rows[i]) all.push_back(kv.first);
    }
    DenseIds docs;
    docs.build(all.begin(), all.end());
    const size_t n_docs = docs.name.size();
    if (n_docs == 0) return;
    // term-major CSR; rows[i] is an ordered map and dense ids follow the same string order, so every
    // term's postings come out ascending by doc id
    std::vector<uint64_t> term_ptr(rows.size() + 1, 0);
    for (size_t i = 0; i < rows.size(); i++) term_ptr[i + 1] = term_ptr[i] + rows[i].size();
    std::vector<uint32_t> post_doc(term_ptr.back());
    std::vector<float> post_tf(term_ptr.back());
    for (size_t i = 0; i < rows.size(); i++) {
        uint64_t j = term_ptr[i];
        for (auto& kv : rows[i]) {
            if (kv.second.empty()) throw std::runtime_error("UpdateTermWeights: posting without a weight entry");
            post_doc[j] = docs.id[kv.first];
            post_tf[j] = kv.second[0];                                  // listPos[0] = normalised tf
            j++;
        }
    }
    std::vector<float> w(post_tf.size());
    std::vector<double> mag(n_docs);
    ss_index* ix = nullptr;
    check(ss_index_create(default_ctx(), n_docs, rows.size(), term_ptr.data(), post_doc.data(), post_tf.data(), &ix), "ss_index_create");
    const int32_t rc = ss_tfidf_build(ix, totalDocs, w.data(), mag.data(), nullptr);   // term_weighting.go:37-44,72
    ss_index_destroy(ix);
    check(rc, "ss_tfidf_build");
    // term_weighting.go:42,47 — weights written back in place, positions untouched
    auto bw = (*inv)->BatchWrite_init(ctx);
    for (size_t i = 0; i < rows.size(); i++) {
        uint64_t j = term_ptr[i];
        for (auto& kv : rows[i]) kv.second[0] = w[j++];
        bw->BatchSet(ctx, comp[i].first, jsonmini::dump(rows[i]));
    }
    bw->Flush(ctx);
    // only docs that occur in this table have an entry in pageMagnitude (:44)
    std::map<std::string, double> pageMagnitude;
    for (uint32_t d : post_doc) pageMagnitude[docs.name[d]] = mag[d];
    saveMagnitude(ctx, pageMagnitude, forw[4], info);
}

}  // namespace ranking
