// minijson.hpp -- the little JSON the driver needs: parse --detector / SHM headers, and write the
// per-frame lines the Zocalo service reads (nlohmann::json::dump(): keys in alphabetical order, no
// spaces -- spotfinder/spotfinder.cc:234-254, 997-1008).
#pragma once
#include <cmath>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ffshost {

struct JsonValue {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::map<std::string, JsonValue> obj;
    bool contains(const std::string& k) const { return kind == OBJ && obj.count(k); }
    const JsonValue& at(const std::string& k) const {
        auto it = obj.find(k);
        if (kind != OBJ || it == obj.end()) throw std::invalid_argument("Key " + k + " is missing from the input JSON");
        return it->second;
    }
    double number() const {
        if (kind == NUM) return num;
        if (kind == STR) return std::stod(str);
        throw std::invalid_argument("JSON value is not a number");
    }
};

class JsonParser {
    const std::string& s;
    size_t i = 0;
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    [[noreturn]] void fail(const char* m) { throw std::invalid_argument(std::string("JSON parse error: ") + m); }
  public:
    explicit JsonParser(const std::string& text) : s(text) {}
    JsonValue parse() {
        JsonValue v = value();
        ws();
        if (i != s.size()) fail("trailing characters");
        return v;
    }
    JsonValue value() {
        ws();
        if (i >= s.size()) fail("unexpected end");
        JsonValue v;
        const char c = s[i];
        if (c == '{') {
            v.kind = JsonValue::OBJ;
            ++i;
            ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            for (;;) {
                ws();
                JsonValue k = value();
                if (k.kind != JsonValue::STR) fail("object key must be a string");
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                ++i;
                v.obj[k.str] = value();
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = JsonValue::ARR;
            ++i;
            ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JsonValue::STR;
            ++i;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) {
                    const char e = s[++i];
                    v.str += e == 'n' ? '\n' : e == 't' ? '\t' : e;
                } else {
                    v.str += s[i];
                }
                ++i;
            }
            if (i >= s.size()) fail("unterminated string");
            ++i;
        } else if (!s.compare(i, 4, "true")) { v.kind = JsonValue::BOOL; v.b = true; i += 4;
        } else if (!s.compare(i, 5, "false")) { v.kind = JsonValue::BOOL; i += 5;
        } else if (!s.compare(i, 4, "null")) { i += 4;
        } else {
            size_t used = 0;
            try { v.num = std::stod(s.substr(i), &used); } catch (...) { fail("bad number"); }
            v.kind = JsonValue::NUM;
            i += used;
        }
        return v;
    }
};

inline std::string json_escape(const std::string& in) {
    std::string o = "\"";
    for (unsigned char c : in) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += (char)c;
    }
    return o + "\"";
}

// float -> shortest text that round-trips, the way nlohmann dumps a float widened to double
inline std::string json_number(double v) {
    if (std::isnan(v) || std::isinf(v)) return "null";
    if (v == std::floor(v) && std::fabs(v) < 1e15) {
        char b[32];
        std::snprintf(b, sizeof b, "%.1f", v);
        return b;
    }
    for (int prec = 1; prec <= 17; ++prec) {
        char b[40];
        std::snprintf(b, sizeof b, "%.*g", prec, v);
        if (std::stod(b) == v) return b;
    }
    return "0";
}

}  // namespace ffshost
