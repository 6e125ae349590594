// Minimal stand-in for dmlc-core's parameter.h (absent submodule), written
// from scratch.  Implements the declarative "struct of typed fields" idiom the
// reference's *Param structs rely on:
//
//   struct P : public dmlc::Parameter<P> {
//     int a; float b;
//     DMLC_DECLARE_PARAMETER(P) {
//       DMLC_DECLARE_FIELD(a).set_default(1).set_range(0, 10);
//       DMLC_DECLARE_FIELD(b);                // required
//     }
//   };
//   DMLC_REGISTER_PARAMETER(P);               // in one .cc
//
// Init*(kwargs) assigns known keys (later duplicates win), applies defaults,
// fails (dmlc::ParamError) on a missing required field, an unparsable value or
// a value outside [lo, hi]; InitAllowUnknown returns the unconsumed pairs in
// their original order.
#ifndef SHIM_DMLC_PARAMETER_H_
#define SHIM_DMLC_PARAMETER_H_
#include <cstddef>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>
#include "./logging.h"

namespace dmlc {

struct ParamError : public dmlc::Error {
  explicit ParamError(const std::string& s) : dmlc::Error(s) {}
};

namespace parameter {

/*! \brief type-erased handle of one declared field */
class FieldAccess {
 public:
  virtual ~FieldAccess() {}
  virtual void Set(void* head, const std::string& value) const = 0;
  virtual void SetDefault(void* head) const = 0;
  virtual void Check(void* head) const = 0;
  virtual std::string Get(const void* head) const = 0;
  bool has_default = false;
  std::string key;
  std::string type;
  std::string description;
};

template <typename T>
inline bool ParseValue(const std::string& s, T* out) {
  std::istringstream is(s);
  is >> *out;
  if (is.fail()) return false;
  // allow trailing whitespace only
  char c;
  while (is.get(c)) if (!isspace(static_cast<unsigned char>(c))) return false;
  return true;
}
template <>
inline bool ParseValue<std::string>(const std::string& s, std::string* out) {
  *out = s;
  return true;
}
template <>
inline bool ParseValue<bool>(const std::string& s, bool* out) {
  if (s == "1" || s == "true" || s == "True") { *out = true; return true; }
  if (s == "0" || s == "false" || s == "False") { *out = false; return true; }
  return false;
}
template <>
inline bool ParseValue<unsigned>(const std::string& s, unsigned* out) {
  long long v;
  if (!ParseValue<long long>(s, &v) || v < 0) return false;
  *out = static_cast<unsigned>(v);
  return true;
}

template <typename T> struct TypeName { static const char* get() { return "value"; } };
template <> struct TypeName<int> { static const char* get() { return "int"; } };
template <> struct TypeName<unsigned> { static const char* get() { return "unsigned"; } };
template <> struct TypeName<float> { static const char* get() { return "float"; } };
template <> struct TypeName<double> { static const char* get() { return "double"; } };
template <> struct TypeName<bool> { static const char* get() { return "boolean"; } };
template <> struct TypeName<std::string> { static const char* get() { return "string"; } };

template <typename T, bool kArith = std::is_arithmetic<T>::value>
struct RangeCheck {
  bool on = false;
  void Set(T, T) {}
  void Run(const std::string&, const T&) const {}
};
template <typename T>
struct RangeCheck<T, true> {
  bool on = false;
  T lo, hi;
  void Set(T a, T b) { on = true; lo = a; hi = b; }
  void Run(const std::string& key, const T& v) const {
    if (on && (v < lo || v > hi)) {
      std::ostringstream os;
      os << "value " << v << " for Parameter " << key << " exceed bound [" << lo << "," << hi << "]";
      throw ParamError(os.str());
    }
  }
};

template <typename T>
class FieldEntry : public FieldAccess {
 public:
  FieldEntry(const std::string& k, ptrdiff_t offset) : offset_(offset) {
    key = k;
    type = TypeName<T>::get();
  }
  // fluent setters used inside DMLC_DECLARE_PARAMETER
  template <typename U>
  FieldEntry& set_default(const U& v) { default_ = static_cast<T>(v); has_default = true; return *this; }
  FieldEntry& set_default(const char* v) { return set_default_str(v); }
  template <typename A, typename B>
  FieldEntry& set_range(A lo, B hi) { range_.Set(static_cast<T>(lo), static_cast<T>(hi)); return *this; }
  template <typename A>
  FieldEntry& set_lower_bound(A lo) {
    range_.Set(static_cast<T>(lo), std::numeric_limits<T>::max());
    return *this;
  }
  FieldEntry& describe(const std::string& d) { description = d; return *this; }

  void Set(void* head, const std::string& value) const override {
    T v;
    if (!ParseValue<T>(value, &v)) {
      throw ParamError("Invalid Parameter format for " + key + " expect " + type + " but value=\'" + value + "\'");
    }
    Ref(head) = v;
  }
  void SetDefault(void* head) const override {
    if (!has_default) {
      throw ParamError("Required parameter " + key + " of " + type + " is not presented");
    }
    Ref(head) = default_;
  }
  void Check(void* head) const override { range_.Run(key, Ref(head)); }
  std::string Get(const void* head) const override {
    std::ostringstream os;
    os << *reinterpret_cast<const T*>(reinterpret_cast<const char*>(head) + offset_);
    return os.str();
  }

 private:
  FieldEntry& set_default_str(const char* v) {
    T t;
    CHECK(ParseValue<T>(v, &t)) << "bad default for " << key;
    default_ = t;
    has_default = true;
    return *this;
  }
  T& Ref(void* head) const { return *reinterpret_cast<T*>(reinterpret_cast<char*>(head) + offset_); }
  ptrdiff_t offset_;
  T default_ = T();
  RangeCheck<T> range_;
};

/*! \brief the ordered field table of one parameter struct */
class ParamManager {
 public:
  void Add(FieldAccess* e) {
    CHECK(index_.count(e->key) == 0) << "key " << e->key << " declared twice in " << name_;
    index_[e->key] = fields_.size();
    fields_.emplace_back(e);
  }
  void set_name(const std::string& n) { name_ = n; }

  template <typename It>
  std::vector<std::pair<std::string, std::string>> Run(void* head, It begin, It end, bool allow_unknown) const {
    std::vector<std::pair<std::string, std::string>> unknown;
    std::vector<char> seen(fields_.size(), 0);
    for (It it = begin; it != end; ++it) {
      auto f = index_.find(it->first);
      if (f == index_.end()) {
        if (!allow_unknown) {
          throw ParamError("Cannot find argument \'" + it->first + "\' in " + name_);
        }
        unknown.push_back(std::make_pair(it->first, it->second));
        continue;
      }
      fields_[f->second]->Set(head, it->second);
      seen[f->second] = 1;
    }
    for (size_t i = 0; i < fields_.size(); ++i) {
      if (!seen[i]) fields_[i]->SetDefault(head);
      fields_[i]->Check(head);
    }
    return unknown;
  }
  std::map<std::string, std::string> Dict(const void* head) const {
    std::map<std::string, std::string> d;
    for (const auto& f : fields_) d[f->key] = f->Get(head);
    return d;
  }

 private:
  std::string name_;
  std::vector<std::unique_ptr<FieldAccess>> fields_;
  std::map<std::string, size_t> index_;
};

template <typename PType>
struct ParamManagerSingleton {
  ParamManager manager;
  explicit ParamManagerSingleton(const std::string& name) {
    PType proto;
    manager.set_name(name);
    proto.__DECLARE__(this);
  }
};

}  // namespace parameter

template <typename PType>
struct Parameter {
 public:
  template <typename Container>
  inline void Init(const Container& kwargs) {
    Guard([&]() { PType::__MANAGER__()->Run(static_cast<PType*>(this), kwargs.begin(), kwargs.end(), false); });
  }
  template <typename Container>
  inline std::vector<std::pair<std::string, std::string>> InitAllowUnknown(const Container& kwargs) {
    std::vector<std::pair<std::string, std::string>> r;
    Guard([&]() { r = PType::__MANAGER__()->Run(static_cast<PType*>(this), kwargs.begin(), kwargs.end(), true); });
    return r;
  }
  inline std::map<std::string, std::string> __DICT__() const {
    return PType::__MANAGER__()->Dict(static_cast<const PType*>(this));
  }

 protected:
  template <typename DType>
  inline parameter::FieldEntry<DType>& DECLARE(parameter::ParamManagerSingleton<PType>* m,
                                               const std::string& key, DType& ref) {
    auto* e = new parameter::FieldEntry<DType>(
        key, reinterpret_cast<char*>(&ref) - reinterpret_cast<char*>(static_cast<PType*>(this)));
    m->manager.Add(e);
    return *e;
  }

 private:
  // with DMLC_LOG_FATAL_THROW=0 (the reference build) a bad config is fatal
  template <typename F>
  static inline void Guard(F f) {
#if DMLC_LOG_FATAL_THROW
    f();
#else
    try {
      f();
    } catch (const ParamError& e) {
      LOG(FATAL) << e.what();
    }
#endif
  }
};

}  // namespace dmlc

#define DMLC_DECLARE_PARAMETER(PType)                       \
  static ::dmlc::parameter::ParamManager* __MANAGER__();    \
  inline void __DECLARE__(::dmlc::parameter::ParamManagerSingleton<PType>* manager)

#define DMLC_DECLARE_FIELD(FieldName) this->DECLARE(manager, #FieldName, FieldName)

#define DMLC_REGISTER_PARAMETER(PType)                                   \
  ::dmlc::parameter::ParamManager* PType::__MANAGER__() {                \
    static ::dmlc::parameter::ParamManagerSingleton<PType> inst(#PType); \
    return &inst.manager;                                                \
  }                                                                      \
  static ::dmlc::parameter::ParamManager* __shim_make_##PType##__ = PType::__MANAGER__()

#endif  // SHIM_DMLC_PARAMETER_H_
