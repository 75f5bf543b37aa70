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

inline void UpdateTopicSensitivePagerank(db::Context& ctx, double dampingFactor, double convergenceCriterion,
                                         std::vector<db::DB*>& forward) {
    using namespace spaghetti;
    std::fprintf(stderr, "Ranking with damping factor='%f', convergence_criteria='%f'\n", dampingFactor, convergenceCriterion);
    // pagerank.go:17-44 — node set = parents U children (frontier pages are nodes without children)
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
    const size_t n = ids.name.size();
    if (n == 0) return;
    std::vector<uint64_t> out_ptr(n + 1, 0);
    for (size_t i = 0; i < nodes.size(); i++) out_ptr[ids.id[nodes[i].first] + 1] = children[i].size();
    for (size_t v = 0; v < n; v++) out_ptr[v + 1] += out_ptr[v];
    std::vector<uint32_t> out_dst(out_ptr[n]);
    for (size_t i = 0; i < nodes.size(); i++) {
        uint64_t base = out_ptr[ids.id[nodes[i].first]];
        for (auto& c : children[i]) out_dst[base++] = ids.id[c];
    }
    // pagerank.go:46-63 — one power iteration per category, differing by numPages only
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
    if (K > 0) {
        ss_graph* g = nullptr;
        check(ss_graph_create(default_ctx(), n, out_dst.size(), out_ptr.data(), out_dst.data(), 0, 1, &g), "ss_graph_create");
        std::vector<int32_t> iters(K);
        const int32_t rc = ss_pagerank_run(g, dampingFactor, convergenceCriterion, 0, K, n_topic.data(), rank.data(), iters.data());
        ss_graph_destroy(g);
        check(rc, "ss_pagerank_run");
    }
    // pagerank.go:65-82 — forw[3][doc] = map[category]rank
    auto bw = forward[3]->BatchWrite_init(ctx);
    for (size_t v = 0; v < n; v++) {
        std::map<std::string, double> PR;
        for (int k = 0; k < K; k++) PR[cat_name[k]] = rank[(size_t)k * n + v];
        bw->BatchSet(ctx, ids.name[v], jsonmini::dump(PR));
    }
    bw->Flush(ctx);
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
