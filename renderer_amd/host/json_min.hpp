// json_min.hpp — a small JSON reader for the glTF loader (objects, arrays, strings with escapes,
// numbers, true / false / null).  Header-only; throws json::ParseError on malformed input (caught
// inside the loader, never crosses its boundary).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace json {

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Value {
    enum class Type { Null, Bool, Number, String, Array, Object } type = Type::Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;  // file order kept (glTF attribute order is observable)

    bool is_null() const { return type == Type::Null; }
    bool is_object() const { return type == Type::Object; }
    bool is_array() const { return type == Type::Array; }
    bool is_number() const { return type == Type::Number; }
    bool is_string() const { return type == Type::String; }
    // member lookup; a missing member is a Null value
    const Value& operator[](const std::string& key) const {
        static const Value null_value;
        if (type == Type::Object)
            for (const auto& kv : obj)
                if (kv.first == key) return kv.second;
        return null_value;
    }
    const Value& operator[](size_t i) const {
        static const Value null_value;
        return (type == Type::Array && i < arr.size()) ? arr[i] : null_value;
    }
    size_t size() const { return type == Type::Array ? arr.size() : (type == Type::Object ? obj.size() : 0); }
    bool has(const std::string& key) const { return !(*this)[key].is_null(); }
    // numbers as indices / counts
    // saturating: a number at or beyond 2^64 (or NaN) must not reach the float -> integer conversion, which is undefined there
    uint64_t as_u64(uint64_t fallback = 0) const {
        if (type != Type::Number || !(num >= 0)) return fallback;
        return num >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)num;
    }
    double as_double(double fallback = 0) const { return type == Type::Number ? num : fallback; }
};

class Parser {
public:
    Parser(const char* p, size_t n) : p_(p), end_(p + n) {}
    Value parse() {
        Value v = value(0);
        ws();
        if (p_ != end_) fail("trailing characters");
        return v;
    }

private:
    const char* p_;
    const char* end_;
    [[noreturn]] void fail(const char* what) { throw ParseError(std::string("JSON: ") + what); }
    void ws() {
        while (p_ != end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_;
    }
    bool eat(char c) {
        ws();
        if (p_ != end_ && *p_ == c) { ++p_; return true; }
        return false;
    }
    void lit(const char* s) {
        for (; *s; ++s, ++p_)
            if (p_ == end_ || *p_ != *s) fail("bad literal");
    }
    Value value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p_ == end_) fail("unexpected end");
        Value v;
        switch (*p_) {
            case '{': {
                ++p_;
                v.type = Value::Type::Object;
                if (eat('}')) return v;
                do {
                    ws();
                    if (p_ == end_ || *p_ != '"') fail("object key expected");
                    std::string k = string();
                    if (!eat(':')) fail("':' expected");
                    v.obj.emplace_back(std::move(k), value(depth + 1));
                } while (eat(','));
                if (!eat('}')) fail("'}' expected");
                return v;
            }
            case '[': {
                ++p_;
                v.type = Value::Type::Array;
                if (eat(']')) return v;
                do v.arr.push_back(value(depth + 1)); while (eat(','));
                if (!eat(']')) fail("']' expected");
                return v;
            }
            case '"': v.type = Value::Type::String; v.str = string(); return v;
            case 't': lit("true"); v.type = Value::Type::Bool; v.b = true; return v;
            case 'f': lit("false"); v.type = Value::Type::Bool; return v;
            case 'n': lit("null"); return v;
            default: {
                const char* s = p_;
                while (p_ != end_ && (*p_ == '-' || *p_ == '+' || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || (*p_ >= '0' && *p_ <= '9'))) ++p_;
                if (s == p_) fail("value expected");
                const std::string t(s, p_);
                char* e = nullptr;
                v.num = std::strtod(t.c_str(), &e);
                if (!e || *e) fail("bad number");
                v.type = Value::Type::Number;
                return v;
            }
        }
    }
    std::string string() {
        ++p_;  // opening quote
        std::string out;
        while (true) {
            if (p_ == end_) fail("unterminated string");
            const char c = *p_++;
            if (c == '"') return out;
            if (c != '\\') { out.push_back(c); continue; }
            if (p_ == end_) fail("bad escape");
            const char e = *p_++;
            switch (e) {
                case '"': case '\\': case '/': out.push_back(e); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    if (end_ - p_ < 4) fail("bad \\u escape");
                    unsigned cp = 0;
                    for (int k = 0; k < 4; ++k) {
                        const char h = *p_++;
                        cp <<= 4;
                        if (h >= '0' && h <= '9') cp |= h - '0';
                        else if (h >= 'a' && h <= 'f') cp |= h - 'a' + 10;
                        else if (h >= 'A' && h <= 'F') cp |= h - 'A' + 10;
                        else fail("bad \\u escape");
                    }
                    // UTF-8 (surrogate pairs are kept as two 3-byte sequences: names only)
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: fail("bad escape");
            }
        }
    }
};

inline Value parse(const char* p, size_t n) { return Parser(p, n).parse(); }

}  // namespace json
