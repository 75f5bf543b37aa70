// json_mini.hpp — the four JSON value shapes of the reference's tables
// (database/noschema_schema.go:125-260), nothing more:
//   []string                      forw[2]  docHash -> children
//   map[string]float64            forw[3], forw[4], forw[5]
//   map[string][]float32          inv[0], inv[1]
// Numbers are written in shortest round-trip form (std::to_chars), like Go's encoding/json, so
// float32 weights and float64 ranks survive storage exactly.
#pragma once
#include <charconv>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace jsonmini {

struct Reader {
    const char* p;
    const char* e;
    explicit Reader(const std::string& s) : p(s.data()), e(s.data() + s.size()) {}
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool peek(char c) { ws(); return p < e && *p == c; }
    void expect(char c) {
        ws();
        if (p >= e || *p != c) throw std::runtime_error(std::string("json: expected '") + c + "'");
        p++;
    }
    bool null_() {
        ws();
        if (e - p >= 4 && std::string(p, 4) == "null") { p += 4; return true; }
        return false;
    }
    std::string str() {
        expect('"');
        std::string out;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= e) break;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {   // keys here are hex hashes / ASCII category names: keep the escape verbatim
                        out += "\\u";
                        break;
                    }
                    default: out += *p;
                }
                p++;
            } else {
                out += *p++;
            }
        }
        expect('"');
        return out;
    }
    double num() {
        ws();
        const char* s = p;
        while (p < e && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) p++;
        if (s == p) throw std::runtime_error("json: number expected");
        double v = 0;
        auto r = std::from_chars(s, p, v);
        if (r.ec != std::errc()) throw std::runtime_error("json: bad number");
        return v;
    }
};

inline std::vector<std::string> parse_string_list(const std::string& s) {
    Reader r(s);
    std::vector<std::string> out;
    if (r.null_()) return out;
    r.expect('[');
    if (r.peek(']')) { r.expect(']'); return out; }
    for (;;) {
        out.push_back(r.str());
        if (r.peek(',')) { r.expect(','); continue; }
        r.expect(']');
        return out;
    }
}

inline std::map<std::string, double> parse_map_f64(const std::string& s) {
    Reader r(s);
    std::map<std::string, double> out;
    if (r.null_()) return out;
    r.expect('{');
    if (r.peek('}')) { r.expect('}'); return out; }
    for (;;) {
        std::string k = r.str();
        r.expect(':');
        out[k] = r.num();
        if (r.peek(',')) { r.expect(','); continue; }
        r.expect('}');
        return out;
    }
}

inline std::map<std::string, std::vector<float>> parse_map_f32list(const std::string& s) {
    Reader r(s);
    std::map<std::string, std::vector<float>> out;
    if (r.null_()) return out;
    r.expect('{');
    if (r.peek('}')) { r.expect('}'); return out; }
    for (;;) {
        std::string k = r.str();
        r.expect(':');
        std::vector<float> v;
        if (!r.null_()) {
            r.expect('[');
            if (!r.peek(']')) {
                for (;;) {
                    v.push_back((float)r.num());     // decimal -> nearest float64 -> float32, as Go's decoder does
                    if (r.peek(',')) { r.expect(','); continue; }
                    break;
                }
            }
            r.expect(']');
        }
        out[k] = std::move(v);
        if (r.peek(',')) { r.expect(','); continue; }
        r.expect('}');
        return out;
    }
}

template <typename T>
inline void put_num(std::string& out, T v) {
    if (v != v || v - v != 0) throw std::runtime_error("json: NaN/Inf is not representable (Go's encoder errors too)");
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);   // shortest round-trip
    out.append(buf, r.ptr);
}
inline void put_str(std::string& out, const std::string& s) {
    out += '"';
    for (char c : s) {
        if (c == '"' || c == '\\') { out += '\\'; out += c; }
        else out += c;
    }
    out += '"';
}
inline std::string dump(const std::map<std::string, double>& m) {
    std::string out = "{";
    bool first = true;
    for (auto& kv : m) {
        if (!first) out += ',';
        first = false;
        put_str(out, kv.first);
        out += ':';
        put_num(out, kv.second);
    }
    return out + "}";
}
inline std::string dump(const std::map<std::string, std::vector<float>>& m) {
    std::string out = "{";
    bool first = true;
    for (auto& kv : m) {
        if (!first) out += ',';
        first = false;
        put_str(out, kv.first);
        out += ":[";
        for (size_t i = 0; i < kv.second.size(); i++) {
            if (i) out += ',';
            put_num(out, kv.second[i]);
        }
        out += ']';
    }
    return out + "}";
}
inline std::string dump(const std::vector<std::string>& v) {
    std::string out = "[";
    for (size_t i = 0; i < v.size(); i++) {
        if (i) out += ',';
        put_str(out, v[i]);
    }
    return out + "]";
}

}  // namespace jsonmini
