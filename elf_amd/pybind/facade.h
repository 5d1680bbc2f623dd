// Batch-collector facade: the Python-visible objects of the reference's `_elf` module
// (src_cpp/elf/Pybind.cc:40-99: Context, SharedMem, SharedMemOptions, AnyP, FuncMapBase, Size,
// ReplyStatus, TSOptions, SearchAlgoOptions) re-designed for a device-resident engine.
//
// The reference's Context is a thread rendezvous: num_games game threads block in sendBatchWait,
// one collector thread per SharedMem gathers <= batchsize states and hands the buffer to the Python
// thread that sits in wait() (elf/base/context.h:184-220,336-353).  Here there are no game threads:
// all games live in HBM and an Engine advances them with kernels until the next batch of network
// inputs is due.  wait() = "run the engine until it has a batch, copy the rows into the tensors the
// caller registered with AnyP.set()", step() = "take the reply rows back".  The protocol the Python
// caller sees (createSharedMemOptions / allocateSharedMem / smem[key].set / wait / step / stop,
// src_py/elf/utils_elf.py:16-103,478-489) is unchanged.
#pragma once
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace elfhip_py {

enum ReplyStatus { SUCCESS = 0, FAILED = 1, UNKNOWN = 2 };  // comm::ReplyStatus (elf/comm/base.h)

// elf::ai::tree_search::SearchAlgoOptions / TSOptions (tree_search_options.h:24-29,91-110); the
// same field names and defaults, because game.py / context_utils.py:91-112 assign them by name.
struct SearchAlgoOptions {
  bool use_prior = true;
  float c_puct = 5;
  bool unexplored_q_zero = false;
  bool root_unexplored_q_zero = false;
};
struct TSOptions {
  int max_num_moves = 0;
  int num_threads = 16;
  int num_rollouts_per_thread = 100;
  int num_rollouts_per_batch = 8;
  bool verbose = true;
  bool verbose_time = false;
  bool persistent_tree = true;
  float root_epsilon = 0.0f;
  float root_alpha = 0.0f;
  std::string pick_method = "most_visited";
  int virtual_loss = 0;
  SearchAlgoOptions alg_opt;
  std::string log_prefix;
  int seed = 0;
};

struct Size {  // elf::Size (elf/base/common.h:41-116)
  std::vector<int> v;
  const std::vector<int>& vec() const { return v; }
  long long nelement() const { long long n = 1; for (int x : v) n *= x; return n; }
};

struct FuncMapBase {  // elf::FuncMapBase (elf/base/extractor.h): name, extents, element type of one key
  std::string name, type_name;
  int type_size = 0;
  int batchsize = 0;
  Size sz;
  int getBatchSize() const { return batchsize; }
  const std::string& getName() const { return name; }
  const Size& getSize() const { return sz; }
  std::string getTypeName() const { return type_name; }
  int getSizeOfType() const { return type_size; }
};

struct AnyP {  // elf::AnyP (extractor.h:345-430): a key bound to caller-owned memory
  FuncMapBase f;
  uint64_t ptr = 0;
  std::vector<long long> strides;  // bytes
  const FuncMapBase* field() const { return &f; }
  std::string info() const;
  void setAddress(uint64_t p, const std::vector<int>& byte_strides);
  long long row_bytes() const { return f.sz.v.empty() ? 0 : f.sz.nelement() / f.sz.v[0] * f.type_size; }
  char* row(int i) const { return (char*)(uintptr_t)ptr + (long long)i * strides.at(0); }
};

struct SharedMemOptions {  // elf::SharedMemOptions (sharedmem.h:26-110)
  int idx = -1;
  std::string label;
  int batchsize = 0;
  int timeout_usec = 0;
  int getIdx() const { return idx; }
  int getBatchSize() const { return batchsize; }
  const std::string& getLabel() const { return label; }
  void setTimeout(int usec) { timeout_usec = usec; }
};

struct SharedMem {  // elf::SharedMem (sharedmem.h:132-294)
  SharedMemOptions opts;
  std::map<std::string, AnyP> mem;
  int effective_batchsize = 0;
  AnyP* get(const std::string& key) {
    auto it = mem.find(key);
    return it == mem.end() ? nullptr : &it->second;
  }
  const SharedMemOptions& getSharedMemOptions() const { return opts; }
  int getEffectiveBatchSize() const { return effective_batchsize; }
  std::string info() const;
};

class Context;

// What a game module plugs into the Context.
class Engine {
 public:
  virtual ~Engine() {}
  // element type / per-sample extents of a batch key (GameFeature::registerExtractor,
  // common/GameFeature.h:166-222); throws for an unknown key
  virtual void describe_key(const std::string& key, FuncMapBase* out) const = 0;
  virtual void on_allocate(SharedMem* smem) = 0;
  virtual void start() = 0;
  // run until a batch is due; nullptr = nothing more will ever come (stopped)
  virtual SharedMem* next_batch() = 0;
  virtual void reply(SharedMem* smem, bool success) = 0;
  virtual void stop() = 0;
};

class Context {  // elf::Context (elf/base/context.h:223-420)
 public:
  explicit Context(Engine* e) : eng_(e) {}
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  SharedMemOptions createSharedMemOptions(const std::string& name, int batchsize) const {
    SharedMemOptions o;
    o.label = name;
    o.batchsize = batchsize;
    return o;
  }
  SharedMem& allocateSharedMem(const SharedMemOptions& o, const std::vector<std::string>& keys);
  void start();
  SharedMem* wait(int timeout_usec);
  void step(ReplyStatus success);
  void stop();
  std::string version() const { return "elfhip-mi355x-abi2"; }
  bool started() const { return started_; }

 private:
  Engine* eng_;
  std::deque<std::unique_ptr<SharedMem>> smems_;
  SharedMem* current_ = nullptr;
  bool started_ = false, stopped_ = false;
};

// ---- inline implementations (each module that includes this gets its own copy) -----------------
inline std::string AnyP::info() const {
  std::string s = "[" + f.name + "][" + f.type_name + "] sz=(";
  for (size_t i = 0; i < f.sz.v.size(); ++i) s += (i ? "," : "") + std::to_string(f.sz.v[i]);
  s += ") strides=(";
  for (size_t i = 0; i < strides.size(); ++i) s += (i ? "," : "") + std::to_string(strides[i]);
  return s + ")" + (ptr ? "" : " unbound");
}

inline void AnyP::setAddress(uint64_t p, const std::vector<int>& byte_strides) {
  if (byte_strides.size() != f.sz.v.size())
    throw std::invalid_argument("AnyP.set(" + f.name + "): " + std::to_string(byte_strides.size()) + " strides for " +
                                std::to_string(f.sz.v.size()) + " dimensions");
  // rows (dimension 0) may be strided; everything inside one row must be dense, as the tensors
  // Allocator._alloc creates are (utils_elf.py:42-50)
  long long expect = f.type_size;
  for (size_t k = f.sz.v.size(); k-- > 1;) {
    if (byte_strides[k] != expect && f.sz.v[k] != 1)
      throw std::invalid_argument("AnyP.set(" + f.name + "): inner dimensions must be contiguous");
    expect *= f.sz.v[k];
  }
  if (!byte_strides.empty() && byte_strides[0] < expect && f.sz.v[0] > 1)
    throw std::invalid_argument("AnyP.set(" + f.name + "): row stride smaller than a row");
  ptr = p;
  strides.assign(byte_strides.begin(), byte_strides.end());
}

inline std::string SharedMem::info() const {
  std::string s = "SharedMem[" + opts.label + "] idx=" + std::to_string(opts.idx) + " batchsize=" +
                  std::to_string(opts.batchsize) + " effective=" + std::to_string(effective_batchsize) + "\n";
  for (auto& kv : mem) s += "  " + kv.second.info() + "\n";
  return s;
}

inline SharedMem& Context::allocateSharedMem(const SharedMemOptions& o, const std::vector<std::string>& keys) {
  if (started_) throw std::runtime_error("Context.allocateSharedMem: context already started");
  if (o.batchsize <= 0) throw std::invalid_argument("Context.allocateSharedMem: batchsize must be positive");
  std::unique_ptr<SharedMem> m(new SharedMem());
  m->opts = o;
  m->opts.idx = (int)smems_.size();
  for (auto& k : keys) {
    AnyP p;
    eng_->describe_key(k, &p.f);
    p.f.name = k;
    p.f.batchsize = o.batchsize;
    p.f.sz.v.insert(p.f.sz.v.begin(), o.batchsize);
    m->mem.emplace(k, std::move(p));
  }
  eng_->on_allocate(m.get());
  smems_.push_back(std::move(m));
  return *smems_.back();
}

inline void Context::start() {
  if (started_) throw std::runtime_error("Context.start: already started");
  for (auto& m : smems_)
    for (auto& kv : m->mem)
      if (!kv.second.ptr) throw std::runtime_error("Context.start: key '" + kv.first + "' of '" + m->opts.label + "' has no memory (AnyP.set)");
  eng_->start();
  started_ = true;
}

inline SharedMem* Context::wait(int) {
  if (!started_) throw std::runtime_error("Context.wait: start() has not been called");
  if (stopped_) return nullptr;
  if (current_) throw std::runtime_error("Context.wait: the previous batch has not been released with step()");
  current_ = eng_->next_batch();
  return current_;
}

inline void Context::step(ReplyStatus s) {
  if (!current_) throw std::runtime_error("Context.step: no batch is out (call wait() first)");
  SharedMem* m = current_;
  current_ = nullptr;
  eng_->reply(m, s == SUCCESS);
}

inline void Context::stop() {
  if (stopped_) return;
  stopped_ = true;
  current_ = nullptr;
  if (started_) eng_->stop();
}

}  // namespace elfhip_py
