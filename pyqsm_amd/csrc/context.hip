// context.hip — library/device management half of the C-ABI (include/pyqsm_hip.h).
#include "common.hpp"

#include <dlfcn.h>

namespace pyqsm {

// ---- roctx ranges --------------------------------------------------------
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
const Roctx& roctx() {
  static Roctx r;
  return r;
}
}  // namespace

ApiRange::ApiRange(const char* name) : on_(roctx().push != nullptr) {
  if (on_) roctx().push(name);
}
ApiRange::~ApiRange() {
  if (on_) roctx().pop();
}

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- arena ------------------------------------------------------------

static constexpr size_t kAlign = 256;

int Arena::alloc(size_t bytes, void** out) {
  bytes = (bytes + kAlign - 1) / kAlign * kAlign;
  if (bytes == 0) bytes = kAlign;
  // bump inside the current chunk, else move on to a later (empty) chunk that is large enough
  for (; cur_ < chunks_.size(); ++cur_) {
    Chunk& c = chunks_[cur_];
    if (c.used + bytes <= c.size) {
      *out = c.base + c.used;
      c.used += bytes;
      return 0;
    }
    if (cur_ + 1 == chunks_.size()) break;
  }
  size_t want = bytes;
  if (want < (size_t(64) << 20)) want = size_t(64) << 20;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess && want > bytes) {
    want = bytes;
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess)
    return fail(PYQSM_ENOMEM, "scratch arena: hipMalloc(%zu) failed: %s", want,
                hipGetErrorString(e));
  chunks_.push_back({static_cast<char*>(p), want, bytes});
  cur_ = chunks_.size() - 1;
  *out = p;
  return 0;
}

int Arena::reset() {
  if (chunks_.size() > 1) {
    size_t total = 0;
    for (auto& c : chunks_) {
      total += c.size;
      (void)hipFree(c.base);
    }
    chunks_.clear();
    void* p = nullptr;
    if (hipMalloc(&p, total) == hipSuccess) chunks_.push_back({static_cast<char*>(p), total, 0});
  } else if (chunks_.size() == 1) {
    chunks_[0].used = 0;
  }
  cur_ = 0;
  return 0;
}

Arena::Mark Arena::mark() const {
  if (chunks_.empty()) return Mark{0, 0};
  return Mark{cur_, chunks_[cur_].used};
}

void Arena::rewind(const Mark& m) {
  if (chunks_.empty()) return;
  for (size_t i = m.chunk + 1; i < chunks_.size(); ++i) chunks_[i].used = 0;
  if (m.chunk < chunks_.size()) {
    chunks_[m.chunk].used = m.used;
    cur_ = m.chunk;
  }
}

void Arena::destroy() {
  for (auto& c : chunks_) (void)hipFree(c.base);
  chunks_.clear();
  cur_ = 0;
}

// ---- page-locked output buffers ----------------------------------------------------------
// Page-locking is slow (~10 ms per 100 MB), so released buffers are kept and handed out again:
// the contraction loop asks for the same three sizes every step. At most kPinnedIdleCap bytes
// sit idle; beyond that, and for small requests, plain malloc.
static constexpr size_t kPinnedMin = size_t(1) << 20;
static constexpr size_t kPinnedIdleCap = size_t(2) << 30;
static std::mutex g_pin_mu;
static std::multimap<size_t, void*>& pin_idle() {
  static auto* m = new std::multimap<size_t, void*>();
  return *m;
}
static std::map<void*, size_t>& pin_live() {
  static auto* m = new std::map<void*, size_t>();
  return *m;
}
static size_t g_pin_idle_bytes = 0;

void* out_alloc(size_t bytes) {
  if (bytes >= kPinnedMin) {
    const size_t want = (bytes + kPinnedMin - 1) / kPinnedMin * kPinnedMin;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = pin_idle().lower_bound(want);
    if (it != pin_idle().end() && it->first <= want + want / 4) {  // at most 25 % larger than asked
      void* p = it->second;
      g_pin_idle_bytes -= it->first;
      pin_live()[p] = it->first;
      pin_idle().erase(it);
      return p;
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess && p) {
      pin_live()[p] = want;
      return p;
    }
    (void)hipGetLastError();  // not fatal: pageable memory does the job
  }
  return malloc(bytes ? bytes : 1);
}

void out_free(void* p) {
  if (!p) return;
  bool pinned = false, unpin = false;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = pin_live().find(p);
    if (it != pin_live().end()) {
      pinned = true;
      const size_t sz = it->second;
      pin_live().erase(it);
      if (g_pin_idle_bytes + sz <= kPinnedIdleCap) {
        pin_idle().emplace(sz, p);
        g_pin_idle_bytes += sz;
      } else {
        unpin = true;
      }
    }
  }
  // unpinning hundreds of megabytes takes a tenth of a second: not under the lock every other
  // thread's allocations wait on
  if (unpin) (void)hipHostFree(p);
  if (!pinned) free(p);
}

// ---- contexts -----------------------------------------------------------

// One context (stream + scratch arena + timers) per device AND calling thread, so that
// host threads can drive independent work on one GPU concurrently: small clouds are
// launch- and sync-latency bound (a 50 k-point tree keeps a few CUs busy), and N trees on
// N threads overlap on the device instead of queueing behind one mutex. A thread keeps its
// contexts until it exits; they then go back to an idle list and are handed to the next
// new thread, so a thread pool that comes and goes does not grow the arenas.
static std::mutex g_mu;
struct Pool {
  std::vector<Ctx*> all, idle;
};
static std::map<int, Pool>& pools() {
  static auto* p = new std::map<int, Pool>();  // never destroyed: threads may outlive exit()
  return *p;
}
static unsigned g_generation = 0;  // bumped by pyqsm_shutdown: held pointers are stale after it

struct ThreadHeld {
  unsigned generation = 0;
  std::map<int, Ctx*> ctx;
  ~ThreadHeld() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (generation != g_generation) return;
    for (auto& kv : ctx) pools()[kv.first].idle.push_back(kv.second);
  }
};
static thread_local ThreadHeld t_held;

static void drain_timers(Ctx* c);

Ctx* ctx_for(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (t_held.generation != g_generation) {
    t_held.ctx.clear();
    t_held.generation = g_generation;
  }
  auto it = t_held.ctx.find(device);
  if (it != t_held.ctx.end()) {
    (void)hipSetDevice(device);
    return it->second;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range (have %d)", device, n);
    return nullptr;
  }
  if ((e = hipSetDevice(device)) != hipSuccess) {
    set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return nullptr;
  }
  Pool& pool = pools()[device];
  if (!pool.idle.empty()) {
    Ctx* c = pool.idle.back();
    pool.idle.pop_back();
    {  // a recycled context must not report the previous thread's launches
      std::lock_guard<std::mutex> lc(c->mu);
      drain_timers(c);
      c->timers.clear();
      c->prof = 0;
    }
    t_held.ctx[device] = c;
    return c;
  }
  Ctx* c = new Ctx();
  c->device = device;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    set_error("hipStreamCreate: %s", hipGetErrorString(e));
    delete c;
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
  pool.all.push_back(c);
  t_held.ctx[device] = c;
  return c;
}

// ---- timers ---------------------------------------------------------------

ProfScope::ProfScope(Ctx* c, const char* name, int weight, int level) : c_(c) {
  if (c->prof < level) return;
  t_ = &c->timers[name];
  hipEvent_t a = nullptr, b = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
    t_ = nullptr;
    return;
  }
  start_ = a;
  (void)hipEventRecord(a, c->stream);
  t_->pending.emplace_back(a, b);
  t_->weights.push_back(weight);
}

ProfScope::~ProfScope() {
  if (!t_) return;
  (void)hipEventRecord(t_->pending.back().second, c_->stream);
}

static void drain_timers(Ctx* c) {
  (void)hipStreamSynchronize(c->stream);
  for (auto& kv : c->timers) {
    Timer& t = kv.second;
    for (size_t i = 0; i < t.pending.size(); ++i) {
      auto& ev = t.pending[i];
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
        t.ms += ms;
        t.launches += t.weights[i];
      }
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    t.pending.clear();
    t.weights.clear();
  }
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pyqsm_init(int device) { return ctx_for(device) ? 0 : PYQSM_ENODEV; }

int pyqsm_shutdown(void) {
  comm_shutdown();
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : pools()) {
    for (Ctx* c : kv.second.all) {
      {
        // a call still in flight on another thread holds c->mu: wait for it to leave before
        // the arena and the stream go away. Callers must not START calls concurrently with
        // pyqsm_shutdown (their thread-local pointers are stale once it returns).
        std::lock_guard<std::mutex> lc(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        drain_timers(c);
        c->arena.destroy();
        (void)hipStreamDestroy(c->stream);
      }
      delete c;
    }
  }
  pools().clear();
  ++g_generation;
  {
    std::lock_guard<std::mutex> lk2(g_pin_mu);
    for (auto& kv : pin_idle()) (void)hipHostFree(kv.second);
    pin_idle().clear();
    g_pin_idle_bytes = 0;
  }
  return 0;
}

const char* pyqsm_last_error(void) { return g_err; }

const char* pyqsm_version(void) { return "pyqsm_hip 0.1.0 gfx950"; }

int pyqsm_sync(int device) {
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

void* pyqsm_stream(int device) {
  Ctx* c = ctx_for(device);
  return c ? static_cast<void*>(c->stream) : nullptr;
}

int pyqsm_dev_malloc(int device, size_t bytes, void** out) {
  if (!out) return fail(PYQSM_EINVAL, "pyqsm_dev_malloc: out is NULL");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  PQ_HIP(hipMalloc(out, bytes ? bytes : 1));
  return 0;
}

int pyqsm_dev_free(int device, void* p) {
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  if (p) PQ_HIP(hipFree(p));
  return 0;
}

int pyqsm_h2d(int device, void* dst, const void* src, size_t bytes) {
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  if (bytes == 0) return 0;
  if (!dst || !src) return fail(PYQSM_EINVAL, "pyqsm_h2d: NULL pointer");
  PQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int pyqsm_d2h(int device, void* dst, const void* src, size_t bytes) {
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  if (bytes == 0) return 0;
  if (!dst || !src) return fail(PYQSM_EINVAL, "pyqsm_d2h: NULL pointer");
  PQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

void pyqsm_free(void* p) { out_free(p); }

void* pyqsm_host_alloc(size_t bytes) { return out_alloc(bytes); }

int pyqsm_prof_enable(int device, int on) {
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->prof = on < 0 ? 0 : on;
  return 0;
}

int pyqsm_prof_reset(int device) {
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  drain_timers(c);
  c->timers.clear();
  return 0;
}

int pyqsm_prof_get(int device, const char* name, double* total_ms, int64_t* launches) {
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  if (!name) return fail(PYQSM_EINVAL, "pyqsm_prof_get: name is NULL");
  std::lock_guard<std::mutex> lk(c->mu);
  drain_timers(c);
  auto it = c->timers.find(name);
  if (total_ms) *total_ms = it == c->timers.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == c->timers.end() ? 0 : it->second.launches;
  return 0;
}

}  // extern "C"
