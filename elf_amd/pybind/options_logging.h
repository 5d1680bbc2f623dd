// `_elf._options` (OptionSpec, OptionMap) and `_elf._logging` (LoggerLevel, Logger, the spdlog-named factory
// functions, getIndexedLogger): the two submodules the reference's Python package star-imports and subclasses
// (src_py/elf/options/py_option_spec.py:25, py_option_map.py:13, src_py/elf/logging/__init__.py,
// configuration.py:12-31, rlpytorch/model_loader.py:192-263).  Same class, method and overload names as
// src_cpp/elf/options/OptionSpec.cc:44-69, OptionMap.cc:21-30, src_cpp/elf/logging/Pybind.cc:37-63, Levels.cc:21-36,
// IndexedLoggerFactory.cc:19-52.  The reference stores option values as nlohmann::json and logs through spdlog;
// neither library is in this image, so values are Python objects (de)serialised with Python's own json module -
// the Python side only ever json.loads what these return - and the logger is a small level-filtered stream writer.
#pragma once
#include <pybind11/pybind11.h>
#include <pybind11/functional.h>
#include <pybind11/stl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <ctime>
#include <functional>
#include <iomanip>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace elfhip_py {
namespace py = pybind11;

// ---- options -----------------------------------------------------------------------------------------------
struct OptionInfo {  // option_spec_detail::OptionBaseTyped<T> (OptionSpec.h:103-182)
  std::string name, help, pytype;   // pytype: "int" | "float" | "bool" | "str" (PythonTypename)
  std::string ctype;                // the C++ type the option was declared with (checkType's identity)
  bool is_list = false, has_default = false;
  py::object def;
};

class OptionSpec {  // OptionSpec.h:228-350
 public:
  OptionSpec() = default;
  OptionSpec(const OptionSpec&) = default;
  bool add(const std::string& name, const std::string& help, const std::string& pytype, const std::string& ctype, bool is_list,
           bool has_default, py::object def) {
    auto o = std::make_shared<OptionInfo>();
    o->name = name; o->help = help; o->pytype = pytype; o->ctype = ctype; o->is_list = is_list; o->has_default = has_default;
    o->def = std::move(def);
    return map_.emplace(name, std::move(o)).second;   // an existing name keeps its first definition
  }
  std::vector<std::string> getOptionNames() const {
    std::vector<std::string> n;
    for (auto& kv : map_) n.push_back(kv.first);
    return n;
  }
  // [{"args": ["--name"], "kwargs": {type, help, required, dest, default?, nargs?, action?}}]  (:164-182, :193-226)
  py::list argparseOptions() const {
    py::list arr;
    for (auto& kv : map_) {
      const OptionInfo& o = *kv.second;
      py::list args;
      py::dict kw;
      kw["help"] = o.help; kw["dest"] = o.name;
      if (o.pytype == "bool" && !o.is_list) {
        const bool flip = o.has_default && py::cast<bool>(o.def);
        kw["action"] = flip ? "store_false" : "store_true";
        args.append((flip ? "--no_" : "--") + o.name);
      } else {
        args.append("--" + o.name);
        kw["type"] = o.pytype; kw["required"] = !o.has_default;
        if (o.has_default) kw["default"] = o.def;
        if (o.is_list) kw["nargs"] = "*";
      }
      py::dict item;
      item["args"] = args; item["kwargs"] = kw;
      arr.append(item);
    }
    return arr;
  }
  std::string getPythonArgparseOptionsAsJSONString() const { return dumps(argparseOptions()); }
  void merge(const OptionSpec& other) { map_.insert(other.map_.begin(), other.map_.end()); }
  void addPrefixSuffixToOptionNames(const std::string& prefix, const std::string& suffix) {
    std::map<std::string, std::shared_ptr<OptionInfo>> m2;
    for (auto& kv : map_) {   // the option objects are shared with every spec they were merged into, as in the reference
      kv.second->name = prefix + kv.second->name + suffix;
      m2.emplace(prefix + kv.first + suffix, kv.second);
    }
    map_ = std::move(m2);
  }
  const OptionInfo& info(const std::string& name) const {
    auto it = map_.find(name);
    if (it == map_.end()) throw std::runtime_error("No option with name " + name + "!");
    return *it->second;
  }
  static std::string dumps(const py::object& o) {
    return py::cast<std::string>(py::module_::import("json").attr("dumps")(o, py::arg("sort_keys") = true,
                                                                            py::arg("separators") = py::make_tuple(",", ":")));
  }
  static py::object loads(const std::string& s) { return py::module_::import("json").attr("loads")(s); }

 private:
  std::map<std::string, std::shared_ptr<OptionInfo>> map_;
};

class OptionMap {  // OptionMap.h:30-108, OptionMap.cc:33-45
 public:
  explicit OptionMap(OptionSpec spec) : spec_(std::move(spec)) {}
  OptionMap(const OptionMap& o) : spec_(o.spec_), data_(py::dict(py::module_::import("copy").attr("deepcopy")(o.data_))) {}
  const OptionSpec& getOptionSpec() const { return spec_; }
  std::string getJSONString() const { return OptionSpec::dumps(data_); }
  void loadJSONString(const std::string& s) {
    py::object d = OptionSpec::loads(s);
    if (!py::isinstance<py::dict>(d)) throw std::runtime_error("loadJSONString: not a JSON object");
    for (auto kv : py::cast<py::dict>(d)) data_[kv.first] = kv.second;
  }
  std::string getAsJSONString(const std::string& name) const {
    if (!data_.contains(name.c_str())) throw std::runtime_error(name + " has not been set!");
    return OptionSpec::dumps(data_[name.c_str()]);
  }
  void setAsJSONString(const std::string& name, const std::string& s) { data_[name.c_str()] = OptionSpec::loads(s); }

 private:
  OptionSpec spec_;
  py::dict data_;
};

template <class T>
void def_add_option(py::class_<OptionSpec>& c, const char* fname, const char* lname, const char* pytype, const char* ctype) {
  c.def(fname, [=](OptionSpec& s, std::string n, std::string h) { return s.add(n, h, pytype, ctype, false, false, py::none()); });
  c.def(fname, [=](OptionSpec& s, std::string n, std::string h, T d) { return s.add(n, h, pytype, ctype, false, true, py::cast(d)); });
  c.def(lname, [=](OptionSpec& s, std::string n, std::string h) { return s.add(n, h, pytype, ctype, true, false, py::none()); });
  c.def(lname, [=](OptionSpec& s, std::string n, std::string h, std::vector<T> d) { return s.add(n, h, pytype, ctype, true, true, py::cast(d)); });
}

inline void register_options(py::module_& m) {
  py::class_<OptionSpec> spec(m, "OptionSpec");
  def_add_option<int32_t>(spec, "addIntOption", "addIntListOption", "int", "int32_t");
  def_add_option<int32_t>(spec, "addInt32Option", "addInt32ListOption", "int", "int32_t");
  def_add_option<uint32_t>(spec, "addUnsignedInt32Option", "addUnsignedInt32ListOption", "int", "uint32_t");
  def_add_option<int64_t>(spec, "addInt64Option", "addInt64ListOption", "int", "int64_t");
  def_add_option<uint64_t>(spec, "addUnsignedInt64Option", "addUnsignedInt64ListOption", "int", "uint64_t");
  def_add_option<double>(spec, "addFloatOption", "addFloatListOption", "float", "double");
  def_add_option<double>(spec, "addFloat64Option", "addFloat64ListOption", "float", "double");
  def_add_option<float>(spec, "addFloat32Option", "addFloat32ListOption", "float", "float");
  def_add_option<bool>(spec, "addBoolOption", "addBoolListOption", "bool", "bool");
  def_add_option<std::string>(spec, "addStrOption", "addStrListOption", "str", "std::string");
  spec.def(py::init<>())
      .def(py::init<const OptionSpec&>())
      .def("getOptionNames", &OptionSpec::getOptionNames)
      .def("getPythonArgparseOptionsAsJSONString", &OptionSpec::getPythonArgparseOptionsAsJSONString)
      .def("merge", &OptionSpec::merge)
      .def("addPrefixSuffixToOptionNames", &OptionSpec::addPrefixSuffixToOptionNames);
  py::class_<OptionMap>(m, "OptionMap")
      .def(py::init<OptionSpec>())
      .def(py::init<const OptionMap&>())
      .def("getOptionSpec", &OptionMap::getOptionSpec)
      .def("getJSONString", &OptionMap::getJSONString)
      .def("loadJSONString", &OptionMap::loadJSONString)
      .def("getAsJSONString", &OptionMap::getAsJSONString)
      .def("setAsJSONString", &OptionMap::setAsJSONString);
}

// ---- logging -----------------------------------------------------------------------------------------------
enum LoggerLevel { L_TRACE = 0, L_DEBUG = 1, L_INFO = 2, L_WARN = 3, L_ERR = 4, L_CRITICAL = 5, L_OFF = 6, L_INVALID = 127 };

class Logger {  // the part of spdlog::logger the reference exposes (logging/Pybind.cc:37-51)
 public:
  Logger(std::string name, FILE* out, bool color) : name_(std::move(name)), out_(out), color_(color), level_(global_level().load()) {}
  void log(LoggerLevel lv, const std::string& msg) {
    if (!should_log(lv)) return;
    static const char* names[] = {"trace", "debug", "info", "warning", "error", "critical", "off"};
    static const char* colors[] = {"\033[37m", "\033[36m", "\033[32m", "\033[33m\033[1m", "\033[31m\033[1m", "\033[1m\033[41m", ""};
    const auto now = std::chrono::system_clock::now();
    const std::time_t t = std::chrono::system_clock::to_time_t(now);
    const int ms = (int)(std::chrono::duration_cast<std::chrono::milliseconds>(now.time_since_epoch()).count() % 1000);
    std::tm tm{};
    localtime_r(&t, &tm);
    char ts[40];
    std::strftime(ts, sizeof(ts), "%Y-%m-%d %H:%M:%S", &tm);
    std::lock_guard<std::mutex> lk(mu());
    if (color_) std::fprintf(out_, "[%s.%03d] [%s] [%s%s\033[0m] %s\n", ts, ms, name_.c_str(), colors[lv], names[lv], msg.c_str());
    else std::fprintf(out_, "[%s.%03d] [%s] [%s] %s\n", ts, ms, name_.c_str(), names[lv], msg.c_str());
    if (lv >= flush_) std::fflush(out_);
  }
  void trace(const std::string& m) { log(L_TRACE, m); }
  void debug(const std::string& m) { log(L_DEBUG, m); }
  void info(const std::string& m) { log(L_INFO, m); }
  void warn(const std::string& m) { log(L_WARN, m); }
  void error(const std::string& m) { log(L_ERR, m); }
  void critical(const std::string& m) { log(L_CRITICAL, m); }
  void flush() { std::fflush(out_); }
  void flush_on(LoggerLevel lv) { flush_ = lv; }
  LoggerLevel level() const { return (LoggerLevel)level_.load(); }
  const std::string& name() const { return name_; }
  void set_level(LoggerLevel lv) { level_ = (int)lv; }
  bool should_log(LoggerLevel lv) const { return lv != L_OFF && (int)lv >= level_.load(); }
  static std::atomic<int>& global_level() { static std::atomic<int> g{(int)L_INFO}; return g; }
  static std::mutex& mu() { static std::mutex m; return m; }

 private:
  std::string name_;
  FILE* out_;
  bool color_;
  std::atomic<int> level_;
  LoggerLevel flush_ = L_OFF;
};

struct LoggerRegistry {  // spdlog's registry: named loggers, creation fails on a duplicate name
  std::map<std::string, std::shared_ptr<Logger>> m;
  std::mutex mu;
  static LoggerRegistry& get() { static LoggerRegistry r; return r; }
  std::shared_ptr<Logger> create(const std::string& name, FILE* out, bool color) {
    std::lock_guard<std::mutex> lk(mu);
    if (m.count(name)) throw std::runtime_error("logger with name '" + name + "' already exists");
    auto l = std::make_shared<Logger>(name, out, color);
    m[name] = l;
    return l;
  }
};

using LoggerCreator = std::function<std::shared_ptr<Logger>(const std::string&)>;
class IndexedLoggerFactory {  // logging/IndexedLoggerFactory.h:59-76, .cc:30-39
 public:
  explicit IndexedLoggerFactory(LoggerCreator c, size_t init = 0) : creator_(std::move(c)), counter_(init) {}
  std::shared_ptr<Logger> makeLogger(const std::string& prefix, const std::string& suffix) {
    const size_t cur = counter_++;
    std::stringstream n;
    n << prefix << std::setw(3) << std::left << cur << suffix;
    return creator_(n.str());
  }

 private:
  LoggerCreator creator_;
  std::atomic<size_t> counter_;
};

inline std::shared_ptr<Logger> getIndexedLogger(const std::string& prefix, const std::string& suffix) {  // .cc:41-52
  static IndexedLoggerFactory factory([](const std::string& name) { return LoggerRegistry::get().create(name, stderr, true); });
  return factory.makeLogger(prefix, suffix);
}

inline void register_logging(py::module_& m) {
  py::class_<IndexedLoggerFactory>(m, "IndexedLoggerFactory")
      .def(py::init<LoggerCreator>())
      .def(py::init<LoggerCreator, size_t>())
      .def("makeLogger", &IndexedLoggerFactory::makeLogger);
  m.def("getIndexedLogger", &getIndexedLogger);
  py::enum_<LoggerLevel>(m, "LoggerLevel")   // logging/Levels.cc:21-36 (spdlog::level::level_enum + invalid)
      .value("trace", L_TRACE).value("debug", L_DEBUG).value("info", L_INFO).value("warn", L_WARN).value("err", L_ERR)
      .value("critical", L_CRITICAL).value("off", L_OFF).value("invalid", L_INVALID)
      .def_static("from_str", [](const char* s) {   // spdlog::level::level_names
        static const char* names[] = {"trace", "debug", "info", "warning", "error", "critical", "off"};
        for (int i = 0; i < 7; ++i) if (std::string(names[i]) == s) return (LoggerLevel)i;
        return L_INVALID;
      });
  py::class_<Logger, std::shared_ptr<Logger>>(m, "Logger")
      .def("trace", &Logger::trace).def("debug", &Logger::debug).def("info", &Logger::info).def("warn", &Logger::warn)
      .def("error", &Logger::error).def("critical", &Logger::critical)
      .def("flush", &Logger::flush).def("flush_on", &Logger::flush_on).def("level", &Logger::level).def("name", &Logger::name)
      .def("set_level", &Logger::set_level).def("should_log", &Logger::should_log);
  m.def("drop", [](const std::string& n) { auto& r = LoggerRegistry::get(); std::lock_guard<std::mutex> lk(r.mu); r.m.erase(n); });
  m.def("drop_all", []() { auto& r = LoggerRegistry::get(); std::lock_guard<std::mutex> lk(r.mu); r.m.clear(); });
  m.def("get", [](const std::string& n) -> std::shared_ptr<Logger> {
    auto& r = LoggerRegistry::get(); std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.m.find(n); return it == r.m.end() ? nullptr : it->second;
  });
  m.def("set_level", [](LoggerLevel lv) {   // spdlog::set_level: every registered logger and the default for new ones
    Logger::global_level() = (int)lv;
    auto& r = LoggerRegistry::get(); std::lock_guard<std::mutex> lk(r.mu);
    for (auto& kv : r.m) kv.second->set_level(lv);
  });
  m.def("set_pattern", [](const std::string&) {});   // the line format is fixed here
  m.def("stdout_logger_mt", [](const std::string& n) { return LoggerRegistry::get().create(n, stdout, false); });
  m.def("stderr_logger_mt", [](const std::string& n) { return LoggerRegistry::get().create(n, stderr, false); });
  m.def("stdout_color_mt", [](const std::string& n) { return LoggerRegistry::get().create(n, stdout, true); });
  m.def("stderr_color_mt", [](const std::string& n) { return LoggerRegistry::get().create(n, stderr, true); });
}

}  // namespace elfhip_py
