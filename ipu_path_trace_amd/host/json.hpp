// Tiny JSON reader (objects, arrays, strings, numbers, booleans, null): replaces boost::property_tree
// for nif_metadata.txt (reference src/neural_networks/NifMetaData.cpp:11-71).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace json {

struct Value {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  const Value& at(const std::string& key) const {
    for (auto& kv : obj) if (kv.first == key) return kv.second;
    throw std::runtime_error("No such node (" + key + ")");
  }
  bool has(const std::string& key) const {
    for (auto& kv : obj) if (kv.first == key) return true;
    return false;
  }
};

class Parser {
public:
  explicit Parser(const std::string& s) : s(s), p(0) {}
  Value parse() { Value v = value(); ws(); if (p != s.size()) fail("trailing characters"); return v; }

private:
  const std::string& s;
  std::size_t p;
  [[noreturn]] void fail(const std::string& m) { throw std::runtime_error("JSON parse error at " + std::to_string(p) + ": " + m); }
  void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p; }
  Value value() {
    ws();
    if (p >= s.size()) fail("unexpected end");
    char c = s[p];
    Value v;
    if (c == '{') {
      v.type = Value::Object; ++p; ws();
      if (s[p] == '}') { ++p; return v; }
      while (true) {
        ws(); Value k = string_(); ws();
        if (s[p] != ':') fail("expected ':'");
        ++p;
        v.obj.emplace_back(k.str, value()); ws();
        if (s[p] == ',') { ++p; continue; }
        if (s[p] == '}') { ++p; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.type = Value::Array; ++p; ws();
      if (s[p] == ']') { ++p; return v; }
      while (true) {
        v.arr.push_back(value()); ws();
        if (s[p] == ',') { ++p; continue; }
        if (s[p] == ']') { ++p; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v = string_();
    } else if (s.compare(p, 4, "true") == 0) { v.type = Value::Bool; v.b = true; p += 4; }
    else if (s.compare(p, 5, "false") == 0) { v.type = Value::Bool; v.b = false; p += 5; }
    else if (s.compare(p, 4, "null") == 0) { p += 4; }
    else {
      char* end = nullptr;
      v.num = std::strtod(s.c_str() + p, &end);
      if (end == s.c_str() + p) fail("bad value");
      v.type = Value::Number;
      v.str.assign(s.c_str() + p, (std::size_t)(end - (s.c_str() + p)));
      p = end - s.c_str();
    }
    return v;
  }
  Value string_() {
    if (s[p] != '"') fail("expected string");
    ++p;
    Value v; v.type = Value::String;
    while (p < s.size() && s[p] != '"') {
      if (s[p] == '\\' && p + 1 < s.size()) {
        char e = s[p + 1];
        v.str.push_back(e == 'n' ? '\n' : e == 't' ? '\t' : e);
        p += 2;
      } else v.str.push_back(s[p++]);
    }
    if (p >= s.size()) fail("unterminated string");
    ++p;
    return v;
  }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace json
