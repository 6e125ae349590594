// Minimal stand-in for dmlc-core's config.h: parses "key = value" streams.
// Tokens are separated by whitespace; '=' may or may not be surrounded by
// spaces; '#' starts a comment that runs to end of line; values may be
// double-quoted (supports \" and \n escapes).  Iteration yields the pairs in
// file order (multi-value mode: duplicates are kept).
#ifndef SHIM_DMLC_CONFIG_H_
#define SHIM_DMLC_CONFIG_H_
#include <istream>
#include <string>
#include <utility>
#include <vector>
#include "./logging.h"
namespace dmlc {
class Config {
 public:
  typedef std::pair<std::string, std::string> ConfigEntry;
  typedef std::vector<ConfigEntry>::const_iterator ConfigIterator;

  explicit Config(bool multi_value = false) : multi_(multi_value) {}
  explicit Config(std::istream& is, bool multi_value = false) : multi_(multi_value) { LoadFromStream(is); }

  void Clear() { entries_.clear(); }

  void LoadFromStream(std::istream& is) {
    std::vector<std::string> toks;
    std::string tok;
    enum { kPlain, kQuote, kComment } st = kPlain;
    auto flush = [&]() { if (!tok.empty()) { toks.push_back(tok); tok.clear(); } };
    bool had_quote = false;
    char c;
    while (is.get(c)) {
      if (st == kComment) {
        if (c == '\n') st = kPlain;
        continue;
      }
      if (st == kQuote) {
        if (c == '\\') {
          char n;
          if (is.get(n)) tok.push_back(n == 'n' ? '\n' : n);
        } else if (c == '"') {
          st = kPlain;
          toks.push_back(tok);  // may be empty
          tok.clear();
          had_quote = false;
        } else {
          tok.push_back(c);
        }
        continue;
      }
      if (c == '#') { flush(); st = kComment; }
      else if (c == '"') { flush(); st = kQuote; had_quote = true; }
      else if (c == '=') { flush(); toks.push_back("="); }
      else if (isspace(static_cast<unsigned char>(c))) { flush(); }
      else tok.push_back(c);
    }
    CHECK(!had_quote || st != kQuote) << "unterminated quote in config";
    flush();
    // expect: key = value triples
    size_t i = 0;
    while (i < toks.size()) {
      CHECK(i + 2 < toks.size() && toks[i + 1] == "=" && toks[i] != "=" && toks[i + 2] != "=")
          << "config parse error near token \'" << toks[i] << "\'";
      Insert(toks[i], toks[i + 2]);
      i += 3;
    }
  }

  void SetParam(const std::string& key, const std::string& value) { Insert(key, value); }

  const std::string& GetParam(const std::string& key) const {
    for (size_t i = entries_.size(); i-- > 0;) if (entries_[i].first == key) return entries_[i].second;
    LOG(FATAL) << "key \"" << key << "\" not found in configuration";
    return entries_[0].second;
  }
  ConfigIterator begin() const { return entries_.begin(); }
  ConfigIterator end() const { return entries_.end(); }

 private:
  void Insert(const std::string& k, const std::string& v) {
    if (!multi_) {
      for (auto& e : entries_) if (e.first == k) { e.second = v; return; }
    }
    entries_.push_back(std::make_pair(k, v));
  }
  bool multi_;
  std::vector<ConfigEntry> entries_;
};
}  // namespace dmlc
#endif  // SHIM_DMLC_CONFIG_H_
