// Minimal JSON reader for config.json / speech_tokenizer config (objects, arrays, strings, numbers,
// booleans, null).  Only what the loader needs; no external dependency.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace qtts {

struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::map<std::string, Json> obj;

    bool has(const std::string& k) const { return kind == Obj && obj.count(k) && obj.at(k).kind != Null; }
    const Json& at(const std::string& k) const {
        static const Json null_json;
        if (kind != Obj) return null_json;
        auto it = obj.find(k);
        return it == obj.end() ? null_json : it->second;
    }
    double num_or(const std::string& k, double d) const { return has(k) && at(k).kind == Num ? at(k).num : d; }
    int int_or(const std::string& k, int d) const { return (int)num_or(k, d); }
    bool bool_or(const std::string& k, bool d) const { return has(k) && at(k).kind == Bool ? at(k).b : d; }
    std::string str_or(const std::string& k, const std::string& d) const { return has(k) && at(k).kind == Str ? at(k).str : d; }
    std::vector<int> ints_or(const std::string& k, std::vector<int> d) const {
        if (!has(k) || at(k).kind != Arr) return d;
        std::vector<int> r;
        for (auto& e : at(k).arr) r.push_back((int)e.num);
        return r;
    }
};

class JsonParser {
    const char* p;
    const char* end;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    [[noreturn]] void fail(const char* what) { throw Error(ERR_BAD_ARG, std::string("json: ") + what); }
    Json value() {
        ws();
        if (p >= end) fail("unexpected end");
        Json j;
        if (*p == '{') {
            j.kind = Json::Obj;
            ++p;
            ws();
            if (p < end && *p == '}') { ++p; return j; }
            for (;;) {
                ws();
                if (p >= end || *p != '"') fail("expected key");
                std::string k = string();
                ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                j.obj[k] = value();
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (*p == '[') {
            j.kind = Json::Arr;
            ++p;
            ws();
            if (p < end && *p == ']') { ++p; return j; }
            for (;;) {
                j.arr.push_back(value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (*p == '"') {
            j.kind = Json::Str;
            j.str = string();
        } else if (end - p >= 4 && std::string(p, 4) == "true") {
            j.kind = Json::Bool; j.b = true; p += 4;
        } else if (end - p >= 5 && std::string(p, 5) == "false") {
            j.kind = Json::Bool; j.b = false; p += 5;
        } else if (end - p >= 4 && std::string(p, 4) == "null") {
            p += 4;
        } else {
            char* e = nullptr;
            j.kind = Json::Num;
            j.num = std::strtod(p, &e);
            if (e == p) fail("bad number");
            p = e;
        }
        return j;
    }
    std::string string() {
        ++p;  // opening quote
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'r': s += '\r'; break;
                    case 'u': s += '?'; p += 4; break;  // non-ASCII escapes are not needed for config keys
                    default: s += *p;
                }
                ++p;
            } else {
                s += *p++;
            }
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return s;
    }

   public:
    static Json parse(const std::string& text) {
        JsonParser ps;
        ps.p = text.data();
        ps.end = text.data() + text.size();
        Json j = ps.value();
        return j;
    }
};

}  // namespace qtts
