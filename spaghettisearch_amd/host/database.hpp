// database.hpp — the slice of the reference's `database` package the ranking path touches
// (database/database.go:42-75 DB interface, batchwriter.go:9-19 BatchWriter), with an in-memory
// table standing in for BadgerDB.  Keys are strings (md5-hex), values are the JSON bytes the
// reference stores (noschema_schema.go:149-204).  Table positions are semantic
// (database.go:85-99): inv[0] title, inv[1] body, forw[2] children, forw[3] ranks,
// forw[4] magnitudes, forw[5] topic metadata.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace db {

struct Context {};   // context.Context is passed through the reference's hot path but never consulted

struct KeyNotFound : std::runtime_error {
    KeyNotFound() : std::runtime_error("Key not found") {}   // badger.ErrKeyNotFound
};

using KV = std::pair<std::string, std::string>;

class BatchWriter {
public:
    virtual ~BatchWriter() = default;
    virtual void BatchSet(Context& ctx, const std::string& key, const std::string& json_value) = 0;
    virtual void Flush(Context& ctx) = 0;
    virtual void Cancel(Context& ctx) = 0;
};

class DB {
public:
    virtual ~DB() = default;
    virtual std::string Get(Context& ctx, const std::string& key) = 0;       // throws KeyNotFound
    virtual void Set(Context& ctx, const std::string& key, const std::string& json_value) = 0;
    virtual bool Has(Context& ctx, const std::string& key) = 0;
    virtual void Delete(Context& ctx, const std::string& key) = 0;
    virtual std::vector<KV> Iterate(Context& ctx) = 0;                       // Collector.KV, database.go:343-358
    virtual std::unique_ptr<BatchWriter> BatchWrite_init(Context& ctx) = 0;
};

class MemDB : public DB {
public:
    std::map<std::string, std::string> rows;
    std::string Get(Context&, const std::string& key) override {
        auto it = rows.find(key);
        if (it == rows.end()) throw KeyNotFound();
        return it->second;
    }
    void Set(Context&, const std::string& key, const std::string& v) override { rows[key] = v; }
    bool Has(Context&, const std::string& key) override { return rows.count(key) != 0; }
    void Delete(Context&, const std::string& key) override { rows.erase(key); }
    std::vector<KV> Iterate(Context&) override { return std::vector<KV>(rows.begin(), rows.end()); }
    std::unique_ptr<BatchWriter> BatchWrite_init(Context&) override {
        struct W : BatchWriter {
            MemDB* d;
            std::vector<KV> pending;
            explicit W(MemDB* d_) : d(d_) {}
            void BatchSet(Context&, const std::string& k, const std::string& v) override { pending.emplace_back(k, v); }
            void Flush(Context&) override {
                for (auto& kv : pending) d->rows[kv.first] = kv.second;
                pending.clear();
            }
            void Cancel(Context&) override { pending.clear(); }
        };
        return std::unique_ptr<BatchWriter>(new W(this));
    }
};

}  // namespace db
