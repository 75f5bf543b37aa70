// retrieval.hpp — host-side mirror of retrieval.Retrieve (retrieval/main_retrieve.go:15) above the C ABI.
//
// The query-time state lives on the GPU: DeviceIndex::load flattens inv[0]/inv[1] (already weighted by
// UpdateTermWeights), forw[4] magnitudes and forw[3] ranks once; Retrieve then parses the query on the
// host, makes ONE library call and maps the k winners into Rank_combined (util.go:25-36).
// Text normalisation (parser.Laundry: Porter2 stemming + stop words) and result decoration
// (get_metadata.go:79-235) are host-side Go in the reference and stay out of scope: `laundry` is a
// hook (default: lower-cased alphanumeric tokens), decoration fills DocHash/PageRank/FinalRank only.
// Quoted phrases (retrieval/phrase.go) are matched on the device from the positional part of the rows.
#pragma once
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <fstream>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#include "md5.hpp"
#include "ranking.hpp"

namespace retrieval {

struct Rank_combined {          // util.go:25-36 (decoration fields are left to the caller)
    std::string DocHash;
    double PageRank = 0;
    double FinalRank = 0;
    double TitleRank = 0, BodyRank = 0;   // diagnostics: the cosine-normalised parts
};

// util.go:151-160: quoted phrases `".*?"`
inline std::vector<std::string> getPhrase(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while ((i = s.find('"', i)) != std::string::npos) {
        const size_t j = s.find('"', i + 1);
        if (j == std::string::npos) break;
        out.push_back(s.substr(i + 1, j - i - 1));
        i = j + 1;
    }
    return out;
}

// parser.go:177-193 without stemming / stop words (those stay host-side Go)
inline std::vector<std::string> defaultLaundry(const std::string& s) {
    std::vector<std::string> out;
    std::string cur;
    for (unsigned char c : s) {
        if (std::isalnum(c)) cur += (char)std::tolower(c);
        else if (!cur.empty()) { out.push_back(cur); cur.clear(); }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

// computeTopicProbs, retrieval/main_retrieve.go:106-159 — DISABLED in the reference (the call at :43 is commented out and
// :87 passes a nil map, so sqd = 0); restated so that the PageRank blend of get_metadata.go:39-42,69 can be switched on.
//   queryTokenised: md5-hex word hashes as Retrieve makes them (:33-36);
//   metadata = forw[5] rows {numPages, wordCount} (:110); inv[2][word] = map[category]frequency (:120-124) — a word
//   that is not in inv[2] throws db::KeyNotFound, where the reference panics (:120-121);
//   as_written = true reproduces `var probs float64` (:142): the product starts at 0, every probability is 0;
//   as_written = false starts it at 1 (multinomial naive Bayes with max-likelihood estimates, uniform prior 1/K, :148).
inline std::map<std::string, double> computeTopicProbs(db::Context& ctx, std::vector<db::DB*>& inv, std::vector<db::DB*>& forw,
                                                       const std::vector<std::string>& queryTokenised, bool as_written) {
    std::map<std::string, std::map<std::string, double>> metadata;                       // :110
    for (auto& kv : forw[5]->Iterate(ctx)) metadata[kv.first] = jsonmini::parse_map_f64(kv.second);
    std::map<std::string, std::vector<double>> topicTF;                                  // :118
    for (auto& tok : queryTokenised) {
        const std::map<std::string, double> topicFreq = jsonmini::parse_map_f64(inv[2]->Get(ctx, tok));   // :120-124 (throws = panic)
        for (auto& tf : topicFreq) topicTF[tf.first].push_back(tf.second);               // :126-134
    }
    std::map<std::string, double> topicProbs;                                            // :137
    for (auto& md : metadata) {
        auto it = topicTF.find(md.first);
        if (it != topicTF.end()) {
            double probs = as_written ? 0.0 : 1.0;                                       // :142
            auto wc = md.second.find("wordCount");
            const double wordCount = wc == md.second.end() ? 0.0 : wc->second;           // missing key reads as 0 in Go
            for (double tf : it->second) probs *= (tf / wordCount);                      // :143-145
            topicProbs[md.first] = probs / (double)metadata.size();                      // :148
        } else {
            topicProbs[md.first] = 0;                                                    // :150
        }
    }
    return topicProbs;
}

// The query-time tables flattened to dense ids: what DeviceIndex::load decodes from the JSON rows and
// uploads, and what a snapshot file holds — so that a server start does not pay the reference's dominant load cost
// (json.Unmarshal of every posting map, database/noschema_schema.go:125-260) again.
//   file = "SSNAP002" | u64 n_docs, n_terms, K | doc names, term names, categories (u32 length + bytes each) |
//          per table (title, body): u64 P, u64 n_pos | term_ptr u64[T+1] | post_doc u32[P] | post_w f32[P] |
//          pos_ptr u64[P+1] | pos f32[n_pos] | mag f64[n_docs]   |   prior f64[K][n_docs]
struct FlatTable {
    std::vector<uint64_t> term_ptr, pos_ptr;
    std::vector<uint32_t> post_doc;
    std::vector<float> post_w, pos;
    std::vector<double> mag;
};
struct FlatCorpus {
    std::vector<std::string> doc_names, term_names, categories;
    FlatTable title, body;
    std::vector<double> prior;       // [K][n_docs]

    template <typename T>
    static void put(std::ostream& f, const std::vector<T>& v) { f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T))); }
    template <typename T>
    static void get(std::istream& f, std::vector<T>& v, size_t n) {
        v.resize(n);
        f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)));
        if (!f) throw std::runtime_error("snapshot: file truncated");
    }
    static void put_u64(std::ostream& f, uint64_t x) { f.write(reinterpret_cast<const char*>(&x), 8); }
    static uint64_t get_u64(std::istream& f) {
        uint64_t x = 0;
        f.read(reinterpret_cast<char*>(&x), 8);
        if (!f) throw std::runtime_error("snapshot: file truncated");
        return x;
    }
    static void put_names(std::ostream& f, const std::vector<std::string>& v) {
        for (auto& s : v) { const uint32_t n = (uint32_t)s.size(); f.write(reinterpret_cast<const char*>(&n), 4); f.write(s.data(), n); }
    }
    static void get_names(std::istream& f, std::vector<std::string>& v, size_t count) {
        v.resize(count);
        for (auto& s : v) {
            uint32_t n = 0;
            f.read(reinterpret_cast<char*>(&n), 4);
            if (!f || n > (1u << 20)) throw std::runtime_error("snapshot: bad name record");
            s.resize(n);
            f.read(&s[0], n);
            if (!f) throw std::runtime_error("snapshot: file truncated");
        }
    }
    void save(const std::string& path) const {
        std::ofstream f(path, std::ios::binary | std::ios::trunc);
        if (!f) throw std::runtime_error("snapshot: cannot open " + path + " for writing");
        f.write("SSNAP002", 8);
        put_u64(f, doc_names.size());
        put_u64(f, term_names.size());
        put_u64(f, categories.size());
        put_names(f, doc_names);
        put_names(f, term_names);
        put_names(f, categories);
        for (const FlatTable* t : {&title, &body}) {
            put_u64(f, t->post_doc.size());
            put_u64(f, t->pos.size());
            put(f, t->term_ptr); put(f, t->post_doc); put(f, t->post_w); put(f, t->pos_ptr); put(f, t->pos); put(f, t->mag);
        }
        put(f, prior);
        if (!f) throw std::runtime_error("snapshot: write to " + path + " failed");
    }
    void load(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("snapshot: cannot open " + path);
        char magic[8];
        f.read(magic, 8);
        if (!f || std::string(magic, 8) != "SSNAP002") throw std::runtime_error("snapshot: " + path + " is not an SSNAP002 file");
        const uint64_t n = get_u64(f), T = get_u64(f), K = get_u64(f);
        get_names(f, doc_names, n);
        get_names(f, term_names, T);
        get_names(f, categories, K);
        for (FlatTable* t : {&title, &body}) {
            const uint64_t P = get_u64(f), np = get_u64(f);
            get(f, t->term_ptr, T + 1); get(f, t->post_doc, P); get(f, t->post_w, P); get(f, t->pos_ptr, P + 1); get(f, t->pos, np); get(f, t->mag, n);
            if (t->term_ptr[T] != P || t->pos_ptr[P] != np) throw std::runtime_error("snapshot: inconsistent table sizes");
        }
        get(f, prior, K * n);
    }
};

class DeviceIndex {
public:
    spaghetti::DenseIds docs, terms;
    FlatCorpus flat;                 // kept for save_snapshot (host memory; drop with flat = {} when not needed)
    ss_index* title = nullptr;
    ss_index* body = nullptr;
    ss_scorer* scorer = nullptr;
    std::vector<std::string> categories;
    std::function<std::vector<std::string>(const std::string&)> laundry = defaultLaundry;

    ~DeviceIndex() {
        if (scorer) ss_scorer_destroy(scorer);
        if (title) ss_index_destroy(title);
        if (body) ss_index_destroy(body);
    }

    void load(db::Context& ctx, std::vector<db::DB*>& forw, std::vector<db::DB*>& inv) {
        using namespace spaghetti;
        const std::vector<db::KV> ranks = forw[3]->Iterate(ctx);
        std::vector<std::map<std::string, std::vector<float>>> trow, brow;
        std::vector<std::string> all_docs, all_terms;
        for (auto& kv : ranks) all_docs.push_back(kv.first);
        const std::vector<db::KV> tcomp = inv[0]->Iterate(ctx), bcomp = inv[1]->Iterate(ctx);
        for (auto& kv : tcomp) { all_terms.push_back(kv.first); trow.push_back(jsonmini::parse_map_f32list(kv.second)); }
        for (auto& kv : bcomp) { all_terms.push_back(kv.first); brow.push_back(jsonmini::parse_map_f32list(kv.second)); }
        for (auto& r : trow) for (auto& kv : r) all_docs.push_back(kv.first);
        for (auto& r : brow) for (auto& kv : r) all_docs.push_back(kv.first);
        docs.build(all_docs.begin(), all_docs.end());
        terms.build(all_terms.begin(), all_terms.end());
        const size_t n = docs.name.size(), T = terms.name.size();
        auto flatten = [&](const std::vector<db::KV>& comp, std::vector<std::map<std::string, std::vector<float>>>& rows,
                           std::vector<uint64_t>& ptr, std::vector<uint32_t>& doc, std::vector<float>& w,
                           std::vector<uint64_t>& pos_ptr, std::vector<float>& pos) {
            std::vector<const std::map<std::string, std::vector<float>>*> by_term(T, nullptr);
            for (size_t i = 0; i < comp.size(); i++) by_term[terms.id[comp[i].first]] = &rows[i];
            ptr.assign(T + 1, 0);
            for (size_t t = 0; t < T; t++) ptr[t + 1] = ptr[t] + (by_term[t] ? by_term[t]->size() : 0);
            doc.resize(ptr[T]);
            w.resize(ptr[T]);
            pos_ptr.assign(1, 0);
            pos.clear();
            for (size_t t = 0; t < T; t++) {
                if (!by_term[t]) continue;
                uint64_t j = ptr[t];
                for (auto& kv : *by_term[t]) {
                    doc[j] = docs.id[kv.first];
                    w[j] = kv.second.at(0);                                            // first entry = norm_tf*idf (main_retrieve.go:227,236)
                    pos.insert(pos.end(), kv.second.begin() + 1, kv.second.end());     // listPos[1:] = positions (phrase.go:144-146)
                    pos_ptr.push_back(pos.size());
                    j++;
                }
            }
        };
        flatten(tcomp, trow, flat.title.term_ptr, flat.title.post_doc, flat.title.post_w, flat.title.pos_ptr, flat.title.pos);
        flatten(bcomp, brow, flat.body.term_ptr, flat.body.post_doc, flat.body.post_w, flat.body.pos_ptr, flat.body.pos);
        // forw[4]: a missing "title"/"body" key reads as 0 (get_metadata.go:57-58, Q8)
        flat.title.mag.assign(n, 0.0);
        flat.body.mag.assign(n, 0.0);
        for (auto& kv : forw[4]->Iterate(ctx)) {
            auto it = docs.id.find(kv.first);
            if (it == docs.id.end()) continue;
            auto m = jsonmini::parse_map_f64(kv.second);
            flat.title.mag[it->second] = m.count("title") ? m["title"] : 0.0;
            flat.body.mag[it->second] = m.count("body") ? m["body"] : 0.0;
        }
        // forw[3]: ranks per category, for the PageRank blend (get_metadata.go:31-42)
        std::vector<std::string> cat;
        for (auto& kv : ranks) for (auto& c : jsonmini::parse_map_f64(kv.second)) cat.push_back(c.first);
        std::sort(cat.begin(), cat.end());
        cat.erase(std::unique(cat.begin(), cat.end()), cat.end());
        flat.categories = cat;
        const size_t K = cat.size();
        flat.prior.assign(K * n, 0.0);
        for (auto& kv : ranks) {
            const uint32_t d = docs.id[kv.first];
            for (auto& c : jsonmini::parse_map_f64(kv.second)) {
                const size_t k = std::lower_bound(cat.begin(), cat.end(), c.first) - cat.begin();
                flat.prior[k * n + d] = c.second;
            }
        }
        flat.doc_names = docs.name;
        flat.term_names = terms.name;
        upload();
    }

    // flat arrays -> device state (ss_index x2, positions, magnitudes, scorer, prior)
    void upload() {
        using namespace spaghetti;
        const size_t n = flat.doc_names.size(), T = flat.term_names.size();
        check(ss_index_create(default_ctx(), n, T, flat.title.term_ptr.data(), flat.title.post_doc.data(), flat.title.post_w.data(), &title), "ss_index_create(title)");
        check(ss_index_create(default_ctx(), n, T, flat.body.term_ptr.data(), flat.body.post_doc.data(), flat.body.post_w.data(), &body), "ss_index_create(body)");
        check(ss_index_set_positions(title, flat.title.pos_ptr.data(), flat.title.pos.data()), "ss_index_set_positions(title)");
        check(ss_index_set_positions(body, flat.body.pos_ptr.data(), flat.body.pos.data()), "ss_index_set_positions(body)");
        check(ss_index_set_weighted(title, flat.title.mag.data()), "ss_index_set_weighted(title)");
        check(ss_index_set_weighted(body, flat.body.mag.data()), "ss_index_set_weighted(body)");
        check(ss_scorer_create(default_ctx(), title, body, &scorer), "ss_scorer_create");
        categories = flat.categories;
        const size_t K = categories.size();
        if (K > 0 && K <= SS_MAX_TOPICS) check(ss_scorer_set_prior(scorer, (int32_t)K, flat.prior.data()), "ss_scorer_set_prior");
    }

    // On-disk snapshot of the flattened tables with the md5-hex <-> dense-id maps: written once after
    // the offline rank update, read at every server start instead of decoding the JSON tables.
    void save_snapshot(const std::string& path) const { flat.save(path); }
    void load_snapshot(const std::string& path) {
        flat.load(path);
        docs.name = flat.doc_names;
        terms.name = flat.term_names;
        docs.id.clear();
        terms.id.clear();
        for (size_t i = 0; i < docs.name.size(); i++) docs.id[docs.name[i]] = (uint32_t)i;
        for (size_t i = 0; i < terms.name.size(); i++) terms.id[terms.name[i]] = (uint32_t)i;
        upload();
    }

    // A batch of queries in one library call (additive API).  topicProbs: per query
    // category -> probability, or empty (nil map in the shipped reference, main_retrieve.go:88: sqd = 0).
    std::vector<std::vector<Rank_combined>> RetrieveBatch(const std::vector<std::string>& queries, int k = 50,
                                                          const std::vector<std::map<std::string, double>>* topicProbs = nullptr) {
        using namespace spaghetti;
        std::vector<uint32_t> q_ptr{0}, q_terms, p_ptr{0}, p_terms;
        std::vector<int32_t> q_len;
        for (std::string query : queries) {
            // main_retrieve.go:17-36
            const std::vector<std::string> phrases = getPhrase(query);
            for (auto& ph : phrases) {
                const size_t pos = query.find("\"" + ph + "\"");
                if (pos != std::string::npos) query.erase(pos, ph.size() + 2);
            }
            std::string joined;
            for (auto& ph : phrases) joined += ph + " ";
            const std::vector<std::string> queryTokenised = laundry(query), phraseTokenised = laundry(joined);
            for (auto& tok : queryTokenised) {
                auto it = terms.id.find(md5::hex(tok));
                q_terms.push_back(it == terms.id.end() ? SS_UNKNOWN_TERM : it->second);   // ErrKeyNotFound tolerated (:193,:218)
            }
            q_ptr.push_back((uint32_t)q_terms.size());
            // all quoted phrases form ONE phrase (main_retrieve.go:26), matched on the device (retrieval/phrase.go)
            for (auto& tok : phraseTokenised) {
                auto it = terms.id.find(md5::hex(tok));
                p_terms.push_back(it == terms.id.end() ? SS_UNKNOWN_TERM : it->second);
            }
            p_ptr.push_back((uint32_t)p_terms.size());
            q_len.push_back((int32_t)(queryTokenised.size() + phraseTokenised.size()));   // :90
        }
        const int nq = (int)queries.size();
        std::vector<double> probs;
        const size_t K = categories.size();
        if (topicProbs && K) {
            probs.assign((size_t)nq * K, 0.0);
            for (int q = 0; q < nq; q++)
                for (auto& kv : (*topicProbs)[q]) {
                    auto it = std::lower_bound(categories.begin(), categories.end(), kv.first);
                    if (it != categories.end() && *it == kv.first) probs[(size_t)q * K + (it - categories.begin())] = kv.second;
                }
        }
        std::vector<ss_hit> hits((size_t)nq * k);
        std::vector<int32_t> n_hits(nq);
        check(ss_score_topk_phrase(scorer, nq, q_ptr.data(), q_terms.data(), p_ptr.data(), p_terms.data(), q_len.data(),
                                   probs.empty() ? nullptr : probs.data(), k, hits.data(), n_hits.data()), "ss_score_topk_phrase");
        std::vector<std::vector<Rank_combined>> out(nq);
        for (int q = 0; q < nq; q++)
            for (int i = 0; i < n_hits[q]; i++) {
                const ss_hit& h = hits[(size_t)q * k + i];
                Rank_combined r;
                r.DocHash = docs.name[h.doc];
                r.PageRank = h.pagerank;      // get_metadata.go:68
                r.FinalRank = h.final;        // get_metadata.go:69
                r.TitleRank = h.title;
                r.BodyRank = h.body;
                out[q].push_back(r);
            }
        return out;
    }
};

// Request batching in front of the device.  The reference serves every HTTP request in its own
// goroutine (cmd/server/server.go:47) and each runs its own Retrieve; here concurrent callers are collected for at most
// `max_wait` (or until `max_batch` are waiting) and answered by ONE ss_score_topk_phrase call.  Retrieve() blocks its
// caller like the reference's function does and returns that caller's own result.
class RetrieveBatcher {
public:
    RetrieveBatcher(DeviceIndex& di, int k = 50, std::chrono::microseconds max_wait = std::chrono::microseconds(1000), size_t max_batch = 1024)
        : di_(di), k_(k), max_wait_(max_wait), max_batch_(max_batch), worker_([this] { run(); }) {}
    ~RetrieveBatcher() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        worker_.join();
    }
    std::vector<Rank_combined> Retrieve(const std::string& query) {
        std::future<std::vector<Rank_combined>> fut;
        {
            std::lock_guard<std::mutex> lk(mu_);
            pending_.emplace_back(query, std::promise<std::vector<Rank_combined>>());
            fut = pending_.back().second.get_future();
        }
        cv_.notify_all();
        return fut.get();               // rethrows what the batch call threw (the reference panics)
    }
    size_t batches() const { return n_batches_; }
    size_t largest_batch() const { return largest_; }

private:
    void run() {
        for (;;) {
            std::vector<std::pair<std::string, std::promise<std::vector<Rank_combined>>>> batch;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !pending_.empty(); });
                if (stop_ && pending_.empty()) return;
                // first request in: wait a little for company
                const auto deadline = std::chrono::steady_clock::now() + max_wait_;
                cv_.wait_until(lk, deadline, [this] { return stop_ || pending_.size() >= max_batch_; });
                batch.swap(pending_);
            }
            std::vector<std::string> queries;
            for (auto& p : batch) queries.push_back(p.first);
            try {
                auto res = di_.RetrieveBatch(queries, k_);
                for (size_t i = 0; i < batch.size(); i++) batch[i].second.set_value(std::move(res[i]));
            } catch (...) {
                for (auto& p : batch) p.second.set_exception(std::current_exception());
            }
            n_batches_++;
            largest_ = std::max(largest_, batch.size());
        }
    }
    DeviceIndex& di_;
    int k_;
    std::chrono::microseconds max_wait_;
    size_t max_batch_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::pair<std::string, std::promise<std::vector<Rank_combined>>>> pending_;
    bool stop_ = false;
    size_t n_batches_ = 0, largest_ = 0;
    std::thread worker_;
};

// retrieval.Retrieve(query, ctx, forw, inv) []Rank_combined — main_retrieve.go:15; first 50 (:99-103)
inline std::vector<Rank_combined> Retrieve(const std::string& query, db::Context& ctx, std::vector<db::DB*>& forw, std::vector<db::DB*>& inv) {
    static DeviceIndex* dev = nullptr;      // loaded on first use, like the server's long-lived tables
    static std::vector<db::DB*>* loaded_for = nullptr;
    if (!dev || loaded_for != &inv) {
        delete dev;
        dev = new DeviceIndex();
        dev->load(ctx, forw, inv);
        loaded_for = &inv;
    }
    return dev->RetrieveBatch({query}, 50)[0];
}

}  // namespace retrieval
