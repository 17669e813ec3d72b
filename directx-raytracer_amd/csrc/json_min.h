// Minimal JSON reader for .crtscene files (the reference vendors rapidjson for this; this is a from-scratch
// recursive-descent DOM with one optimisation that matters for multi-million-triangle scenes: an array
// whose elements are all numbers is stored as a flat std::vector<double>, not as boxed values).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace crt::json {

struct Value;
using Member = std::pair<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, NumArray, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<double> nums;
    std::vector<Member> obj;

    bool isNull() const { return kind == Null; }
    bool isObject() const { return kind == Object; }
    bool isArray() const { return kind == Array || kind == NumArray; }
    bool isNumber() const { return kind == Number; }
    bool isString() const { return kind == String; }
    bool isBool() const { return kind == Bool; }

    // nullptr when absent (the reference dereferences MemberEnd() for absent keys; we do not)
    const Value* find(const char* key) const
    {
        if (kind != Object) return nullptr;
        for (const Member& m : obj)
            if (m.first == key) return &m.second;
        return nullptr;
    }
    size_t size() const { return kind == NumArray ? nums.size() : arr.size(); }
    // number at index i of an array (either representation)
    double numberAt(size_t i) const
    {
        if (kind == NumArray) return nums[i];
        if (kind == Array && arr[i].kind == Number) return arr[i].num;
        throw std::runtime_error("json: array element is not a number");
    }
};

class Parser {
public:
    explicit Parser(const std::string& text) : p(text.data()), end(text.data() + text.size()), begin(text.data()) {}

    Value parseDocument()
    {
        Value v = parseValue(0);
        skipWs();
        if (p != end) fail("trailing characters");
        return v;
    }

private:
    const char* p;
    const char* end;
    const char* begin;

    [[noreturn]] void fail(const char* what) const
    {
        throw std::runtime_error(std::string("json: ") + what + " at byte " + std::to_string(p - begin));
    }
    void skipWs()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool startsNumber() const { return p < end && (*p == '-' || (*p >= '0' && *p <= '9')); }

    double parseNumber()
    {
        // strtod needs a terminated buffer; numbers are short, copy the candidate span
        char buf[64];
        size_t n = 0;
        const char* q = p;
        while (q < end && n < sizeof(buf) - 1 &&
               ((*q >= '0' && *q <= '9') || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E'))
            buf[n++] = *q++;
        buf[n] = 0;
        char* stop = nullptr;
        const double d = std::strtod(buf, &stop);
        if (stop == buf) fail("bad number");
        p += (stop - buf);
        return d;
    }

    std::string parseString()
    {
        if (*p != '"') fail("expected string");
        p++;
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= end) fail("bad escape");
                switch (*p) {
                case 'n': s += '\n'; break;
                case 't': s += '\t'; break;
                case 'r': s += '\r'; break;
                case 'b': s += '\b'; break;
                case 'f': s += '\f'; break;
                case 'u': { // BMP code point -> UTF-8
                    if (end - p < 5) fail("bad \\u escape");
                    unsigned cp = 0;
                    for (int i = 1; i <= 4; i++) {
                        const char c = p[i];
                        cp <<= 4;
                        if (c >= '0' && c <= '9') cp |= unsigned(c - '0');
                        else if (c >= 'a' && c <= 'f') cp |= unsigned(c - 'a' + 10);
                        else if (c >= 'A' && c <= 'F') cp |= unsigned(c - 'A' + 10);
                        else fail("bad \\u escape");
                    }
                    p += 4;
                    if (cp < 0x80) s += char(cp);
                    else if (cp < 0x800) { s += char(0xC0 | (cp >> 6)); s += char(0x80 | (cp & 0x3F)); }
                    else { s += char(0xE0 | (cp >> 12)); s += char(0x80 | ((cp >> 6) & 0x3F)); s += char(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: s += *p; break; // \" \\ \/
                }
                p++;
            } else {
                s += *p++;
            }
        }
        if (p >= end) fail("unterminated string");
        p++;
        return s;
    }

    Value parseValue(int depth)
    {
        if (depth > 64) fail("nesting too deep");
        skipWs();
        if (p >= end) fail("unexpected end");
        Value v;
        const char c = *p;
        if (c == '{') {
            v.kind = Value::Object;
            p++;
            skipWs();
            if (p < end && *p == '}') { p++; return v; }
            for (;;) {
                skipWs();
                std::string key = parseString();
                skipWs();
                if (p >= end || *p != ':') fail("expected ':'");
                p++;
                v.obj.emplace_back(std::move(key), parseValue(depth + 1));
                skipWs();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            p++;
            skipWs();
            if (p < end && *p == ']') { p++; v.kind = Value::NumArray; return v; }
            if (startsNumber()) {
                // fast path: flat number array; falls back to boxed values if a non-number shows up
                v.kind = Value::NumArray;
                for (;;) {
                    skipWs();
                    if (!startsNumber()) break;
                    v.nums.push_back(parseNumber());
                    skipWs();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == ']') { p++; return v; }
                    fail("expected ',' or ']'");
                }
                v.kind = Value::Array;
                v.arr.reserve(v.nums.size() + 1);
                for (double d : v.nums) { Value n; n.kind = Value::Number; n.num = d; v.arr.push_back(std::move(n)); }
                v.nums.clear();
            } else {
                v.kind = Value::Array;
            }
            for (;;) {
                v.arr.push_back(parseValue(depth + 1));
                skipWs();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = Value::String;
            v.str = parseString();
        } else if (startsNumber()) {
            v.kind = Value::Number;
            v.num = parseNumber();
        } else if (end - p >= 4 && std::memcmp(p, "true", 4) == 0) {
            v.kind = Value::Bool; v.b = true; p += 4;
        } else if (end - p >= 5 && std::memcmp(p, "false", 5) == 0) {
            v.kind = Value::Bool; v.b = false; p += 5;
        } else if (end - p >= 4 && std::memcmp(p, "null", 4) == 0) {
            v.kind = Value::Null; p += 4;
        } else {
            fail("unexpected character");
        }
        return v;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parseDocument(); }

} // namespace crt::json
