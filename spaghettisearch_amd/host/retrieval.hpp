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
#include <deque>
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

// What the indexer writes for one page (indexer/indexer.go:23-348): its own postings in inv[0] / inv[1] (setInverted
// :350-408, listPos = [normTF, positions...]), the anchor words it puts into the TITLE rows of its children (:236-290,
// positions -100) and its forw[2] row.  `before` = what checkAndUpdate (:420-641) removes, `after` = what the re-index adds.
struct PageIndexInfo {
    std::string docHash;
    std::map<std::string, std::vector<float>> title, body;                           // wordHash -> listPos
    std::map<std::string, std::map<std::string, std::vector<float>>> anchors;        // childHash -> wordHash -> listPos
    std::vector<std::string> children;                                               // forw[2][docHash]
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
    bool mags_resident = false;      // squared magnitudes resident on the device: deltas keep them up to date
    bool flat_stale = false;         // deltas were applied since `flat` was filled

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
        if (scorer) { ss_scorer_destroy(scorer); scorer = nullptr; }
        if (title) { ss_index_destroy(title); title = nullptr; }
        if (body) { ss_index_destroy(body); body = nullptr; }
        mags_resident = flat_stale = false;
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
    void save_snapshot(const std::string& path) { sync_flat(); flat.save(path); }
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

    // ---- incremental re-index of one page on the resident tables --------------------------------------
    // The table side repeats the reference's writes: checkAndUpdate removes the page's old title words from inv[0]
    // (indexer.go:455-492), its old body words from inv[1] (:494-531) and the postings its anchor words made in its old
    // children's title rows (:533-616; the child's WHOLE posting of that word goes, also what other parents put there);
    // the re-index upserts the new postings (`value[docHash] = ...`, :395-403 and :277-285) and rewrites forw[2] (:301-304).
    // [The reference collects these writes in batch writers fed from reads of the pre-state, so of two removals that hit
    // the same row only the last (in goroutine order) survives the flush; here every removal is applied.]
    // The device side is ONE delta per table: pairs to delete, postings to add (an upsert deletes its pair first), merged
    // into the resident CSR by ss_index_apply_delta_pos after growing the doc / term space for new children and new words.
    // Weights are stored as given (listPos[0]).  update_magnitudes: the touched docs' magnitudes follow the delta on the
    // device (O(delta) once the squared magnitudes are resident) and their forw[4] rows are rewritten; false leaves
    // forw[4] and the device magnitudes as they were, which is what the reference serves until the next UpdateTermWeights.
    void ApplyDelta(db::Context& ctx, std::vector<db::DB*>& forw, std::vector<db::DB*>& inv, const PageIndexInfo& before,
                    const PageIndexInfo& after, bool update_magnitudes = true) {
        using namespace spaghetti;
        if (!title || !body) throw std::runtime_error("ApplyDelta: the index is not loaded");
        if (before.docHash != after.docHash) throw std::runtime_error("ApplyDelta: before and after describe different pages");
        struct Post { std::string term, doc; std::vector<float> listPos; };
        std::vector<std::pair<std::string, std::string>> del[2];          // (wordHash, docHash) per table
        std::vector<Post> add[2];
        for (auto& kv : before.title) del[0].emplace_back(kv.first, before.docHash);
        for (auto& kv : before.body) del[1].emplace_back(kv.first, before.docHash);
        for (auto& ch : before.anchors)
            for (auto& kv : ch.second) del[0].emplace_back(kv.first, ch.first);
        for (auto& kv : after.title) add[0].push_back({kv.first, after.docHash, kv.second});
        for (auto& kv : after.body) add[1].push_back({kv.first, after.docHash, kv.second});
        for (auto& ch : after.anchors)
            for (auto& kv : ch.second) add[0].push_back({kv.first, ch.first, kv.second});
        for (int t = 0; t < 2; t++)
            for (auto& a : add[t])
                if (a.listPos.empty()) throw std::runtime_error("ApplyDelta: posting without a weight entry");
        // --- device -----------------------------------------------------------------------------------------------------
        auto grow = [](DenseIds& ids, const std::string& h) {
            auto it = ids.id.find(h);
            if (it != ids.id.end()) return it->second;
            const uint32_t v = (uint32_t)ids.name.size();
            ids.name.push_back(h);
            ids.id[h] = v;
            return v;
        };
        // The device goes FIRST and the tables follow only once both resident tables have taken their delta: a failing
        // library call must not leave inv[] / forw[2] ahead of the index that answers queries.  Whatever happens, the index
        // stays servable: the scorer is re-created on every way out (ScorerGuard), and ids grown for a delta that did not
        // happen are taken back while the device tables still have their old size.
        const size_t n_before = docs.name.size(), t_before = terms.name.size();
        for (int t = 0; t < 2; t++)
            for (auto& a : add[t]) { grow(docs, a.doc); grow(terms, a.term); }
        for (auto& c : after.children) grow(docs, c);
        const size_t n = docs.name.size(), T = terms.name.size();
        struct ScorerGuard {
            DeviceIndex& di;
            ~ScorerGuard() {
                if (di.scorer || !di.title || !di.body) return;
                const size_t K = di.categories.size(), nd = di.docs.name.size();
                if (ss_scorer_create(spaghetti::default_ctx(), di.title, di.body, &di.scorer) != SS_OK) { di.scorer = nullptr; return; }
                if (K > 0 && K <= SS_MAX_TOPICS && di.flat.prior.size() == K * nd) (void)ss_scorer_set_prior(di.scorer, (int32_t)K, di.flat.prior.data());
            }
        } scorer_guard{*this};
        bool resized = false;
        auto shrink_ids = [&] {
            if (resized) return;                                          // the device tables hold the new ids (as empty docs / terms)
            for (size_t v = n_before; v < docs.name.size(); v++) docs.id.erase(docs.name[v]);
            docs.name.resize(n_before);
            for (size_t v = t_before; v < terms.name.size(); v++) terms.id.erase(terms.name[v]);
            terms.name.resize(t_before);
        };
        if (scorer) { ss_scorer_destroy(scorer); scorer = nullptr; }      // scorers on a table go before its delta
        try {
            if (n != n_before || T != t_before) {
                check(ss_index_resize(title, n, T), "ss_index_resize(title)");
                resized = true;
                check(ss_index_resize(body, n, T), "ss_index_resize(body)");
            }
        } catch (...) {
            if (resized) (void)ss_index_resize(body, n, T);               // title grew: body must follow or the ids diverge
            shrink_ids();
            throw;
        }
        // the PageRank table follows the doc space at once (the guard's scorer needs a table of the new size): new docs hold 0
        // until the next PageRank run (ReloadPrior)
        if (n != n_before && !categories.empty()) {
            const size_t K = categories.size();
            std::vector<double> grown(K * n, 0.0);
            for (size_t k = 0; k < K; k++) std::copy(flat.prior.begin() + k * n_before, flat.prior.begin() + (k + 1) * n_before, grown.begin() + k * n);
            flat.prior.swap(grown);
        }
        if (update_magnitudes && !mags_resident) {                        // once: squared magnitudes of the stored weights
            check(ss_index_refresh_magnitudes(title, nullptr), "ss_index_refresh_magnitudes(title)");
            check(ss_index_refresh_magnitudes(body, nullptr), "ss_index_refresh_magnitudes(body)");
            mags_resident = true;
        }
        std::vector<uint32_t> touched;
        ss_index* table[2] = {title, body};
        for (int t = 0; t < 2; t++) {
            std::vector<uint32_t> del_term, del_doc, add_term, add_doc;
            std::vector<float> add_w, add_pos;
            std::vector<uint64_t> add_pos_ptr{0};
            std::map<std::pair<uint32_t, uint32_t>, bool> seen;
            auto del_pair = [&](const std::string& w, const std::string& d) {
                auto ti = terms.id.find(w);
                auto di = docs.id.find(d);
                if (ti == terms.id.end() || di == docs.id.end()) return;  // never indexed: nothing to delete
                if (seen.emplace(std::make_pair(ti->second, di->second), true).second) {
                    del_term.push_back(ti->second);
                    del_doc.push_back(di->second);
                    touched.push_back(di->second);
                }
            };
            for (auto& d : del[t]) del_pair(d.first, d.second);
            for (auto& a : add[t]) del_pair(a.term, a.doc);               // upsert
            std::map<std::pair<uint32_t, uint32_t>, const Post*> last;    // the last write of a (word, doc) wins, as in a map
            for (auto& a : add[t]) last[{terms.id[a.term], docs.id[a.doc]}] = &a;
            for (auto& kv : last) {
                add_term.push_back(kv.first.first);
                add_doc.push_back(kv.first.second);
                add_w.push_back(kv.second->listPos[0]);
                add_pos.insert(add_pos.end(), kv.second->listPos.begin() + 1, kv.second->listPos.end());
                add_pos_ptr.push_back(add_pos.size());
                touched.push_back(kv.first.second);
            }
            const int32_t rc = ss_index_apply_delta_pos(table[t], 0, nullptr, del_term.size(), del_term.data(), del_doc.data(), add_term.size(),
                                                        add_term.data(), add_doc.data(), add_w.data(), add_pos_ptr.data(), add_pos.data());
            if (rc != SS_OK) {
                flat_stale = true;
                const std::string why = ss_last_error(default_ctx());
                if (t == 0) { shrink_ids(); throw std::runtime_error("ss_index_apply_delta_pos(title): " + why + " (nothing changed)"); }
                throw std::runtime_error("ss_index_apply_delta_pos(body): " + why + " — the resident title table already holds its delta and the tables do not: load() again");
            }
        }
        // --- tables (the device has taken both deltas) -----------------------------------------------------------------------------------------------------
        for (int t = 0; t < 2; t++) {
            std::map<std::string, std::map<std::string, std::vector<float>>> rows;   // the rows this delta touches
            auto row_of = [&](const std::string& w) -> std::map<std::string, std::vector<float>>& {
                auto it = rows.find(w);
                if (it != rows.end()) return it->second;
                auto& r = rows[w];
                if (inv[t]->Has(ctx, w)) r = jsonmini::parse_map_f32list(inv[t]->Get(ctx, w));
                return r;
            };
            for (auto& d : del[t]) row_of(d.first).erase(d.second);
            for (auto& a : add[t]) row_of(a.term)[a.doc] = a.listPos;
            auto bw = inv[t]->BatchWrite_init(ctx);
            for (auto& r : rows) {
                if (r.second.empty()) { if (inv[t]->Has(ctx, r.first)) inv[t]->Delete(ctx, r.first); }   // "delete this row" (:484-489)
                else bw->BatchSet(ctx, r.first, jsonmini::dump(r.second));
            }
            bw->Flush(ctx);
        }
        forw[2]->Set(ctx, after.docHash, jsonmini::dump(after.children));

        std::sort(touched.begin(), touched.end());
        touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        if (update_magnitudes && !touched.empty()) {
            std::vector<double> mt(touched.size()), mb(touched.size());
            check(ss_index_read_magnitudes(title, touched.size(), touched.data(), mt.data()), "ss_index_read_magnitudes(title)");
            check(ss_index_read_magnitudes(body, touched.size(), touched.data(), mb.data()), "ss_index_read_magnitudes(body)");
            auto bw = forw[4]->BatchWrite_init(ctx);
            for (size_t i = 0; i < touched.size(); i++) {
                const std::string& d = docs.name[touched[i]];
                std::map<std::string, double> row;
                if (forw[4]->Has(ctx, d)) row = jsonmini::parse_map_f64(forw[4]->Get(ctx, d));
                row["title"] = mt[i];
                row["body"] = mb[i];
                bw->BatchSet(ctx, d, jsonmini::dump(row));
            }
            bw->Flush(ctx);
        }
        flat_stale = true;
        const size_t K = categories.size();
        check(ss_scorer_create(default_ctx(), title, body, &scorer), "ss_scorer_create");
        if (K > 0 && K <= SS_MAX_TOPICS) check(ss_scorer_set_prior(scorer, (int32_t)K, flat.prior.data()), "ss_scorer_set_prior");
    }

    // forw[3] was rewritten (UpdateTopicSensitivePagerank / ResidentPagerank::Run): refresh the scorer's PageRank table.
    void ReloadPrior(db::Context& ctx, std::vector<db::DB*>& forw) {
        using namespace spaghetti;
        const std::vector<db::KV> ranks = forw[3]->Iterate(ctx);
        std::vector<std::string> cat;
        for (auto& kv : ranks) for (auto& c : jsonmini::parse_map_f64(kv.second)) cat.push_back(c.first);
        std::sort(cat.begin(), cat.end());
        cat.erase(std::unique(cat.begin(), cat.end()), cat.end());
        const size_t K = cat.size(), n = docs.name.size();
        flat.categories = categories = cat;
        flat.prior.assign(K * n, 0.0);
        for (auto& kv : ranks) {
            auto it = docs.id.find(kv.first);
            if (it == docs.id.end()) continue;                 // a node that is in no table yet: cannot be retrieved either
            for (auto& c : jsonmini::parse_map_f64(kv.second))
                flat.prior[(std::lower_bound(cat.begin(), cat.end(), c.first) - cat.begin()) * n + it->second] = c.second;
        }
        if (scorer) check(ss_scorer_set_prior(scorer, K <= SS_MAX_TOPICS ? (int32_t)K : 0, flat.prior.data()), "ss_scorer_set_prior");
    }

    // after deltas the flattened host copy is read back from the device (only save_snapshot needs it)
    void sync_flat() {
        using namespace spaghetti;
        if (!flat_stale) return;
        const size_t n = docs.name.size();
        flat.doc_names = docs.name;
        flat.term_names = terms.name;
        ss_index* table[2] = {title, body};
        FlatTable* ft[2] = {&flat.title, &flat.body};
        for (int t = 0; t < 2; t++) {
            uint64_t nd = 0, nt = 0, np = 0;
            check(ss_index_get_info(table[t], &nd, &nt, &np), "ss_index_get_info");
            ft[t]->term_ptr.resize(nt + 1);
            ft[t]->post_doc.resize(np);
            ft[t]->post_w.resize(np);
            check(ss_index_read(table[t], ft[t]->term_ptr.data(), ft[t]->post_doc.data(), ft[t]->post_w.data()), "ss_index_read");
            ft[t]->pos_ptr.resize(np + 1);
            check(ss_index_read_positions(table[t], ft[t]->pos_ptr.data(), nullptr), "ss_index_read_positions");
            ft[t]->pos.resize(ft[t]->pos_ptr[np]);
            check(ss_index_read_positions(table[t], nullptr, ft[t]->pos.data()), "ss_index_read_positions");
            std::vector<uint32_t> all(n);
            for (size_t i = 0; i < n; i++) all[i] = (uint32_t)i;
            ft[t]->mag.resize(n);
            check(ss_index_read_magnitudes(table[t], n, all.data(), ft[t]->mag.data()), "ss_index_read_magnitudes");
        }
        flat_stale = false;
    }

    // OPT-IN: the ODP keyword vectors for computeTopicProbs at query time — the call the reference has
    // commented out (main_retrieve.go:41-43,87-88).  Kept as hash maps in host memory: a query has a handful of words.
    std::unordered_map<std::string, std::vector<std::pair<uint32_t, double>>> keyword_topics;   // wordHash -> (index into topic_names, frequency)
    std::vector<std::string> topic_names;                                                       // forw[5] keys
    std::vector<double> topic_word_count;
    void LoadTopics(db::Context& ctx, std::vector<db::DB*>& forw, std::vector<db::DB*>& inv) {
        topic_names.clear();
        topic_word_count.clear();
        keyword_topics.clear();
        for (auto& kv : forw[5]->Iterate(ctx)) {
            auto md = jsonmini::parse_map_f64(kv.second);
            topic_names.push_back(kv.first);
            topic_word_count.push_back(md.count("wordCount") ? md["wordCount"] : 0.0);
        }
        for (auto& kv : inv[2]->Iterate(ctx)) {
            auto& v = keyword_topics[kv.first];
            for (auto& tf : jsonmini::parse_map_f64(kv.second)) {
                auto it = std::lower_bound(topic_names.begin(), topic_names.end(), tf.first);
                if (it != topic_names.end() && *it == tf.first) v.emplace_back((uint32_t)(it - topic_names.begin()), tf.second);
            }
        }
    }
    // computeTopicProbs (main_retrieve.go:106-159) with the product started at 1 (as_written = false above), on the
    // resident keyword vectors.  A query word outside inv[2] carries no topic information and is skipped — the reference
    // would panic on it (:120-121), which no serving path can afford.
    std::map<std::string, double> liveTopicProbs(const std::vector<std::string>& queryTokenised) const {
        const size_t K = topic_names.size();
        std::vector<double> probs(K, 1.0);
        std::vector<char> seen(K, 0);
        for (auto& tok : queryTokenised) {
            auto it = keyword_topics.find(tok);
            if (it == keyword_topics.end()) continue;
            for (auto& cf : it->second) { probs[cf.first] *= cf.second / topic_word_count[cf.first]; seen[cf.first] = 1; }   // :143-145
        }
        std::map<std::string, double> out;
        for (size_t c = 0; c < K; c++) out[topic_names[c]] = seen[c] ? probs[c] / (double)K : 0.0;                            // :148,:150
        return out;
    }

    // A batch of queries in one library call (additive API).  topicProbs: per query
    // category -> probability, or empty (nil map in the shipped reference, main_retrieve.go:88: sqd = 0).
    // live_topic_probs (opt-in, default off = the reference's nil map): every query's probabilities come from
    // liveTopicProbs over its non-phrase words (the argument of the commented-out call, main_retrieve.go:43).
    // what a batch of query strings becomes on its way to the device (main_retrieve.go:17-36,88-90)
    struct Tokenised {
        std::vector<uint32_t> q_ptr{0}, q_terms, p_ptr{0}, p_terms;
        std::vector<int32_t> q_len;
        std::vector<double> probs;
        int nq = 0;
    };
    Tokenised tokenise(const std::vector<std::string>& queries, const std::vector<std::map<std::string, double>>* topicProbs,
                       bool live_topic_probs) const {
        using namespace spaghetti;
        Tokenised t;
        std::vector<std::map<std::string, double>> live;
        if (live_topic_probs) {
            if (topicProbs) throw std::runtime_error("RetrieveBatch: explicit and live topic probabilities are exclusive");
            if (topic_names.empty()) throw std::runtime_error("RetrieveBatch: live topic probabilities need LoadTopics");
        }
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
            std::vector<std::string> hashed;
            for (auto& tok : queryTokenised) {
                hashed.push_back(md5::hex(tok));
                auto it = terms.id.find(hashed.back());
                t.q_terms.push_back(it == terms.id.end() ? SS_UNKNOWN_TERM : it->second);   // ErrKeyNotFound tolerated (:193,:218)
            }
            if (live_topic_probs) live.push_back(liveTopicProbs(hashed));
            t.q_ptr.push_back((uint32_t)t.q_terms.size());
            // all quoted phrases form ONE phrase (main_retrieve.go:26), matched on the device (retrieval/phrase.go)
            for (auto& tok : phraseTokenised) {
                auto it = terms.id.find(md5::hex(tok));
                t.p_terms.push_back(it == terms.id.end() ? SS_UNKNOWN_TERM : it->second);
            }
            t.p_ptr.push_back((uint32_t)t.p_terms.size());
            t.q_len.push_back((int32_t)(queryTokenised.size() + phraseTokenised.size()));   // :90
        }
        t.nq = (int)queries.size();
        const size_t K = categories.size();
        if (live_topic_probs) topicProbs = &live;
        if (topicProbs && K) {
            t.probs.assign((size_t)t.nq * K, 0.0);
            for (int q = 0; q < t.nq; q++)
                for (auto& kv : (*topicProbs)[q]) {
                    auto it = std::lower_bound(categories.begin(), categories.end(), kv.first);
                    if (it != categories.end() && *it == kv.first) t.probs[(size_t)q * K + (it - categories.begin())] = kv.second;
                }
        }
        return t;
    }
    std::vector<std::vector<Rank_combined>> to_ranks(int nq, int k, const std::vector<ss_hit>& hits, const std::vector<int32_t>& n_hits) const {
        std::vector<std::vector<Rank_combined>> out(nq);
        for (int q = 0; q < nq; q++)
            for (int i = 0; i < n_hits[q]; i++) {
                const auto& h = hits[(size_t)q * k + i];
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
    std::vector<std::vector<Rank_combined>> RetrieveBatch(const std::vector<std::string>& queries, int k = 50,
                                                          const std::vector<std::map<std::string, double>>* topicProbs = nullptr,
                                                          bool live_topic_probs = false) {
        using namespace spaghetti;
        const Tokenised t = tokenise(queries, topicProbs, live_topic_probs);
        std::vector<ss_hit> hits((size_t)t.nq * k);
        std::vector<int32_t> n_hits(t.nq);
        check(ss_score_topk_phrase(scorer, t.nq, t.q_ptr.data(), t.q_terms.data(), t.p_ptr.data(), t.p_terms.data(), t.q_len.data(),
                                   t.probs.empty() ? nullptr : t.probs.data(), k, hits.data(), n_hits.data()), "ss_score_topk_phrase");
        return to_ranks(t.nq, k, hits, n_hits);
    }
    // The same in two halves, for a caller that has the next batch ready while this one runs (RetrieveBatcher): BeginBatch tokenises
    // and enqueues (ss_score_topk_submit; up to SS_SCORE_INFLIGHT batches), FinishBatch waits for that batch and converts its rows.
    struct PendingBatch { uint64_t ticket = 0; int nq = 0, k = 0; };
    PendingBatch BeginBatch(const std::vector<std::string>& queries, int k = 50) {
        using namespace spaghetti;
        const Tokenised t = tokenise(queries, nullptr, false);
        PendingBatch pb;
        pb.nq = t.nq;
        pb.k = k;
        check(ss_score_topk_submit(scorer, t.nq, t.q_ptr.data(), t.q_terms.data(), t.p_ptr.data(), t.p_terms.data(), t.q_len.data(), nullptr, k,
                                   &pb.ticket), "ss_score_topk_submit");
        return pb;
    }
    std::vector<std::vector<Rank_combined>> FinishBatch(const PendingBatch& pb) {
        using namespace spaghetti;
        std::vector<ss_hit> hits((size_t)pb.nq * pb.k + 1);
        std::vector<int32_t> n_hits((size_t)pb.nq + 1);
        check(ss_score_topk_collect(scorer, pb.ticket, hits.data(), n_hits.data()), "ss_score_topk_collect");
        return to_ranks(pb.nq, pb.k, hits, n_hits);
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
    typedef std::vector<std::pair<std::string, std::promise<std::vector<Rank_combined>>>> Batch;
    struct InFlight { DeviceIndex::PendingBatch pb; Batch batch; };
    // every caller of `batch` that has no answer yet is served on its own: only the caller of an offending query gets the error
    // (done[i]: caller i has its answer — a promise takes exactly one value or exception: a second set_* throws future_error,
    // which inside a catch block would end the serving thread)
    void one_by_one(Batch& batch, std::vector<char>& done) {
        for (size_t i = 0; i < batch.size(); i++) {
            if (done[i]) continue;
            std::exception_ptr err;
            try {
                auto one = di_.RetrieveBatch({batch[i].first}, k_);
                if (one.size() != 1) throw std::runtime_error("RetrieveBatch: no result for a single query");
                batch[i].second.set_value(std::move(one[0]));
                done[i] = 1;
            } catch (...) {
                err = std::current_exception();
            }
            if (!done[i]) {
                try { batch[i].second.set_exception(err); } catch (...) {}   // promise already satisfied: nothing left to tell
                done[i] = 1;
            }
        }
    }
    void finish(InFlight& f) {
        std::vector<char> done(f.batch.size(), 0);
        bool whole = true;
        try {
            auto res = di_.FinishBatch(f.pb);
            if (res.size() != f.batch.size()) throw std::runtime_error("RetrieveBatch: result count differs from the batch");
            for (size_t i = 0; i < f.batch.size(); i++) {
                f.batch[i].second.set_value(std::move(res[i]));
                done[i] = 1;
            }
        } catch (...) {
            whole = false;
        }
        if (!whole) one_by_one(f.batch, done);
    }
    // Up to TWO batches in flight: while the device scores batch i, this thread tokenises and enqueues batch i+1 (if callers are
    // waiting), then hands batch i back.  With nobody waiting a batch is handed back as soon as it is done, as before.
    void run() {
        std::deque<InFlight> flight;
        for (;;) {
            Batch batch;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (flight.empty()) {
                    cv_.wait(lk, [this] { return stop_ || !pending_.empty(); });
                    if (stop_ && pending_.empty()) return;
                    // first request in: wait a little for company
                    const auto deadline = std::chrono::steady_clock::now() + max_wait_;
                    cv_.wait_until(lk, deadline, [this] { return stop_ || pending_.size() >= max_batch_; });
                }
                batch.swap(pending_);                                      // (a batch is running: whoever waits now rides the next one)
            }
            if (!batch.empty()) {
                std::vector<std::string> queries;
                for (auto& p : batch) queries.push_back(p.first);
                n_batches_++;
                largest_ = std::max(largest_, batch.size());
                InFlight f;
                bool begun = true;
                try {
                    f.pb = di_.BeginBatch(queries, k_);
                } catch (...) {
                    begun = false;                                         // the library refuses the batch as a whole (e.g. one phrase beyond SS_MAX_PHRASE_TERMS)
                }
                if (begun) {
                    f.batch = std::move(batch);
                    flight.push_back(std::move(f));
                } else {
                    std::vector<char> done(batch.size(), 0);
                    one_by_one(batch, done);
                }
            }
            // hand back the oldest batch when a second one is behind it, or when nobody is waiting to be batched
            bool more;
            { std::lock_guard<std::mutex> lk(mu_); more = !pending_.empty(); }
            while (!flight.empty() && (flight.size() >= 2 || !more)) {
                finish(flight.front());
                flight.pop_front();
            }
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
