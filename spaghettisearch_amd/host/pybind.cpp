// pybind.cpp — Python binding of the C++ host layer (module spaghettisearch_amd._host), so that the
// pytest parity tests can drive the reference-shaped entry points:
//   ranking.UpdateTopicSensitivePagerank / ranking.UpdateTermWeights / retrieval.Retrieve
// over in-memory tables holding the reference's JSON row formats.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "retrieval.hpp"

namespace py = pybind11;

PYBIND11_MODULE(_host, m) {
    m.doc() = "C++ host mirror of SpaghettiSearch's ranking/retrieval entry points above the HIP C ABI";
    py::class_<db::MemDB>(m, "MemDB")
        .def(py::init<>())
        .def("set", [](db::MemDB& d, const std::string& k, const std::string& v) { d.rows[k] = v; })
        .def("get", [](db::MemDB& d, const std::string& k) {
            auto it = d.rows.find(k);
            if (it == d.rows.end()) throw py::key_error(k);
            return it->second;
        })
        .def("keys", [](db::MemDB& d) {
            std::vector<std::string> out;
            for (auto& kv : d.rows) out.push_back(kv.first);
            return out;
        })
        .def("__len__", [](db::MemDB& d) { return d.rows.size(); })
        .def("clear", [](db::MemDB& d) { d.rows.clear(); });

    py::class_<retrieval::Rank_combined>(m, "Rank_combined")
        .def_readonly("DocHash", &retrieval::Rank_combined::DocHash)
        .def_readonly("PageRank", &retrieval::Rank_combined::PageRank)
        .def_readonly("FinalRank", &retrieval::Rank_combined::FinalRank)
        .def_readonly("TitleRank", &retrieval::Rank_combined::TitleRank)
        .def_readonly("BodyRank", &retrieval::Rank_combined::BodyRank);

    auto as_dbs = [](std::vector<db::MemDB*>& v) {
        std::vector<db::DB*> out;
        for (auto* p : v) out.push_back(p);
        return out;
    };
    m.def("UpdateTopicSensitivePagerank", [as_dbs](double d, double eps, std::vector<db::MemDB*> forward, bool teleport_sets,
                                                   std::vector<db::MemDB*> inv, bool two_vectors) {
        db::Context ctx;
        auto f = as_dbs(forward), i = as_dbs(inv);
        ranking::TopicSensitive ts;
        ts.teleport_sets = teleport_sets;
        ts.inv = &i;
        ts.two_vectors = two_vectors;
        ranking::UpdateTopicSensitivePagerank(ctx, d, eps, f, ts);
    }, py::arg("dampingFactor"), py::arg("convergenceCriterion"), py::arg("forward"), py::arg("teleport_sets") = false,
       py::arg("inv") = std::vector<db::MemDB*>(), py::arg("two_vectors") = false);
    m.def("TopicTeleportSets", [as_dbs](std::vector<db::MemDB*> forward, std::vector<db::MemDB*> inv) {
        db::Context ctx;
        auto f = as_dbs(forward), i = as_dbs(inv);
        return ranking::TopicTeleportSets(ctx, f, i);
    });
    py::class_<ranking::ResidentPagerank>(m, "ResidentPagerank")
        .def(py::init<>())
        .def("Build", [as_dbs](ranking::ResidentPagerank& r, std::vector<db::MemDB*> forward) {
            db::Context ctx;
            auto f = as_dbs(forward);
            r.Build(ctx, f);
        })
        .def("ApplyDelta", [as_dbs](ranking::ResidentPagerank& r, std::vector<db::MemDB*> forward, const std::vector<std::string>& parents) {
            db::Context ctx;
            auto f = as_dbs(forward);
            r.ApplyDelta(ctx, f, parents);
        })
        .def("Run", [as_dbs](ranking::ResidentPagerank& r, double d, double eps, std::vector<db::MemDB*> forward) {
            db::Context ctx;
            auto f = as_dbs(forward);
            r.Run(ctx, d, eps, f);
        })
        .def_readonly("name", &ranking::ResidentPagerank::name)
        .def_readonly("rebuilds", &ranking::ResidentPagerank::rebuilds);
    auto page_info = [](const py::dict& d) {
        retrieval::PageIndexInfo p;
        p.docHash = d["docHash"].cast<std::string>();
        if (d.contains("title")) p.title = d["title"].cast<std::map<std::string, std::vector<float>>>();
        if (d.contains("body")) p.body = d["body"].cast<std::map<std::string, std::vector<float>>>();
        if (d.contains("anchors")) p.anchors = d["anchors"].cast<std::map<std::string, std::map<std::string, std::vector<float>>>>();
        if (d.contains("children")) p.children = d["children"].cast<std::vector<std::string>>();
        return p;
    };
    m.def("UpdateTermWeights", [as_dbs](db::MemDB* inv, std::vector<db::MemDB*> forw, const std::string& info) {
        db::Context ctx;
        auto f = as_dbs(forw);
        db::DB* i = inv;
        ranking::UpdateTermWeights(ctx, &i, f, info);
    }, py::arg("inv"), py::arg("forw"), py::arg("info"));
    m.def("md5_hex", &md5::hex);
    m.def("computeTopicProbs", [as_dbs](std::vector<db::MemDB*> inv, std::vector<db::MemDB*> forw, const std::vector<std::string>& tokens, bool as_written) {
        db::Context ctx;
        auto i = as_dbs(inv), f = as_dbs(forw);
        try {
            return retrieval::computeTopicProbs(ctx, i, f, tokens, as_written);
        } catch (const db::KeyNotFound&) {
            throw py::key_error("query word not in inv[2] (the reference panics, main_retrieve.go:120-121)");
        }
    }, py::arg("inv"), py::arg("forw"), py::arg("queryTokenised"), py::arg("as_written") = true);

    py::class_<retrieval::DeviceIndex>(m, "DeviceIndex")
        .def(py::init<>())
        .def("load", [as_dbs](retrieval::DeviceIndex& di, std::vector<db::MemDB*> forw, std::vector<db::MemDB*> inv) {
            db::Context ctx;
            auto f = as_dbs(forw), i = as_dbs(inv);
            di.load(ctx, f, i);
        })
        .def("RetrieveBatch", [](retrieval::DeviceIndex& di, const std::vector<std::string>& queries, int k,
                                 py::object topic_probs, bool live_topic_probs) {
            if (topic_probs.is_none()) return di.RetrieveBatch(queries, k, nullptr, live_topic_probs);
            auto tp = topic_probs.cast<std::vector<std::map<std::string, double>>>();
            return di.RetrieveBatch(queries, k, &tp, live_topic_probs);
        }, py::arg("queries"), py::arg("k") = 50, py::arg("topic_probs") = py::none(), py::arg("live_topic_probs") = false)
        .def("LoadTopics", [as_dbs](retrieval::DeviceIndex& di, std::vector<db::MemDB*> forw, std::vector<db::MemDB*> inv) {
            db::Context ctx;
            auto f = as_dbs(forw), i = as_dbs(inv);
            di.LoadTopics(ctx, f, i);
        })
        .def("liveTopicProbs", &retrieval::DeviceIndex::liveTopicProbs)
        .def("ApplyDelta", [as_dbs, page_info](retrieval::DeviceIndex& di, std::vector<db::MemDB*> forw, std::vector<db::MemDB*> inv,
                                               const py::dict& before, const py::dict& after, bool update_magnitudes) {
            db::Context ctx;
            auto f = as_dbs(forw), i = as_dbs(inv);
            di.ApplyDelta(ctx, f, i, page_info(before), page_info(after), update_magnitudes);
        }, py::arg("forw"), py::arg("inv"), py::arg("before"), py::arg("after"), py::arg("update_magnitudes") = true)
        .def("ReloadPrior", [as_dbs](retrieval::DeviceIndex& di, std::vector<db::MemDB*> forw) {
            db::Context ctx;
            auto f = as_dbs(forw);
            di.ReloadPrior(ctx, f);
        })
        .def("save_snapshot", &retrieval::DeviceIndex::save_snapshot)
        .def("load_snapshot", &retrieval::DeviceIndex::load_snapshot)
        .def_readonly("categories", &retrieval::DeviceIndex::categories);
    py::class_<retrieval::RetrieveBatcher>(m, "RetrieveBatcher")
        .def(py::init([](retrieval::DeviceIndex& di, int k, int max_wait_us, size_t max_batch) {
                 return new retrieval::RetrieveBatcher(di, k, std::chrono::microseconds(max_wait_us), max_batch);
             }), py::arg("index"), py::arg("k") = 50, py::arg("max_wait_us") = 1000, py::arg("max_batch") = 1024, py::keep_alive<1, 2>())
        .def("Retrieve", [](retrieval::RetrieveBatcher& b, const std::string& q) {
            py::gil_scoped_release rel;          // other Python threads queue their requests meanwhile
            return b.Retrieve(q);
        })
        .def_property_readonly("batches", &retrieval::RetrieveBatcher::batches)
        .def_property_readonly("largest_batch", &retrieval::RetrieveBatcher::largest_batch);
    m.def("Retrieve", [as_dbs](const std::string& query, std::vector<db::MemDB*> forw, std::vector<db::MemDB*> inv) {
        db::Context ctx;
        auto f = as_dbs(forw), i = as_dbs(inv);
        retrieval::DeviceIndex di;          // the binding keeps no global state: load per call
        di.load(ctx, f, i);
        return di.RetrieveBatch({query}, 50)[0];
    });
}
