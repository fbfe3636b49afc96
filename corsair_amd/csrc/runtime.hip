// Library runtime: error string, scratch pool, per-kernel-family profiling.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include <atomic>

namespace cs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- pool -------------------------------------------------------------------------------
// Freed blocks are cached PER HOST THREAD: a thread drives one stream, so a block it frees while its
// kernels are still queued is handed out again only to later work of the same stream (stream order
// makes that safe without a synchronisation).  Another thread, on another stream, never receives it:
// every live block remembers the thread that allocated it, and a block freed by a DIFFERENT thread
// (a map handle dropped by Python's garbage collector on a helper or worker thread, say) does not enter
// that thread's cache -- it goes back to the driver with hipFree, which waits for the device, so work the
// owner's stream still has in flight on the block finishes first.  Rare and slow by design.
static std::mutex g_pool_mu;
struct LiveBlock {
  size_t cls;                  // size class
  unsigned long long owner;    // id of the allocating thread (monotonic: the ADDRESS of a thread-local cache can
                               // be handed to a later thread once its owner has exited)
};
static std::map<void*, LiveBlock> g_live;             // block -> size class + owner (all threads)
static unsigned long long g_foreign_frees = 0;        // blocks freed by a thread other than their owner
static std::atomic<unsigned long long> g_next_thread_id{1};
struct ThreadCache {
  const unsigned long long id = g_next_thread_id.fetch_add(1, std::memory_order_relaxed);
  std::map<size_t, std::vector<void*>> free_;          // size class -> blocks
  ~ThreadCache() {
    for (auto& kv : free_)
      for (void* p : kv.second) (void)hipFree(p);      // fails harmlessly after runtime shutdown
  }
};
static thread_local ThreadCache t_cache;
// pool_defer_begin .. pool_defer_end: blocks freed in between change hands only at the end (a call that runs its kernels on
// several streams joins them on the caller's stream first: the cache is ordered against ONE stream)
static thread_local bool t_defer = false;
static thread_local std::vector<std::pair<size_t, void*>> t_deferred;
static thread_local hipStream_t t_stream = nullptr;
static thread_local bool t_stream_set = false;

// Entry points that free scratch without waiting for their kernels call this first: the cache is only
// stream-ordered, so a thread that switches streams drains the old one before blocks change hands.
static void download_reset();
void pool_use_stream(hipStream_t s) {
  download_reset();  // entry of a public call: downloads left pending by a failed call are dropped
  if (t_stream_set && t_stream != s) (void)hipStreamSynchronize(t_stream);
  t_stream = s;
  t_stream_set = true;
}

static size_t size_class(size_t bytes) {
  size_t c = 256;
  while (c < bytes) c <<= 1;
  // above 64 MiB round to 16 MiB multiples instead of powers of two (288 GB HBM is large, but
  // doubling multi-GB tables is still wasteful)
  if (c > (64u << 20)) {
    const size_t g = 16u << 20;
    c = (bytes + g - 1) / g * g;
  }
  return c;
}

static void trim_thread_cache() {
  for (auto& kv : t_cache.free_) {
    for (void* p : kv.second) (void)hipFree(p);
    kv.second.clear();
  }
}

void* pool_alloc(size_t bytes) {
  size_t c = size_class(bytes);
  void* p = nullptr;
  auto it = t_cache.free_.find(c);
  if (it != t_cache.free_.end() && !it->second.empty()) {
    p = it->second.back();
    it->second.pop_back();
  }
  if (!p) {
    hipError_t e = hipMalloc(&p, c);
    if (e != hipSuccess) {
      trim_thread_cache();
      e = hipMalloc(&p, c);
      if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", c, hipGetErrorString(e));
        return nullptr;
      }
    }
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_live[p] = LiveBlock{c, t_cache.id};
  return p;
}

void pool_free(void* p) {
  if (!p) return;
  LiveBlock b;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;
    b = it->second;
    g_live.erase(it);
    if (b.owner != t_cache.id) ++g_foreign_frees;
  }
  if (b.owner != t_cache.id) {
    // not ours: the owner's stream may still be reading it and this thread's stream is not ordered
    // against that stream.  hipFree synchronises the device before it releases the memory.
    const hipError_t e = hipFree(p);
    if (e != hipSuccess) set_error("pool_free: hipFree of a block of another thread failed: %s", hipGetErrorString(e));
    return;
  }
  // ours: later work of this thread is enqueued behind whatever still uses the block (same stream, or a
  // stream this thread drained when it switched: pool_use_stream)
  if (t_defer)
    t_deferred.emplace_back(b.cls, p);
  else
    t_cache.free_[b.cls].push_back(p);
}

void pool_defer_begin() { t_defer = true; }
void pool_defer_end() {
  t_defer = false;
  for (const auto& d : t_deferred) t_cache.free_[d.first].push_back(d.second);
  t_deferred.clear();
}

void pool_stats(unsigned long long out[3]) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  out[0] = (unsigned long long)g_live.size();
  out[1] = g_foreign_frees;
  unsigned long long cached = 0;
  for (const auto& kv : t_cache.free_) cached += (unsigned long long)kv.first * kv.second.size();
  out[2] = cached;
}

// ---- small device -> host downloads ----------------------------------------------------------
// hipMemcpyAsync into pageable host memory makes the runtime block on the stream first (slow wake-up,
// tens to hundreds of microseconds of GPU idle after a long kernel) and stage the bytes afterwards.
// The sizes, counters and flags the host needs go through a page-locked buffer of the calling thread
// instead: download_async enqueues, download_sync waits once and hands the bytes out.
namespace {
struct Downloads {
  char* pinned = nullptr;
  size_t cap = 0, used = 0;
  struct Item {
    void* dst;
    size_t off, n;
  };
  std::vector<Item> items;
  ~Downloads() {
    if (pinned) (void)hipHostFree(pinned);
  }
};
thread_local Downloads t_dl;
}  // namespace

static void download_reset() {
  t_dl.items.clear();
  t_dl.used = 0;
}

hipError_t download_async(void* host_dst, const void* dev_src, size_t bytes, hipStream_t s) {
  const size_t need = t_dl.used + ((bytes + 15) / 16) * 16;
  if (need > t_dl.cap) {
    if (!t_dl.items.empty()) {  // cannot move a buffer with copies in flight: fall back to a plain copy
      hipError_t e = hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s);
      return e;
    }
    size_t cap = t_dl.cap ? t_dl.cap : 4096;
    while (cap < need) cap *= 2;
    if (t_dl.pinned) (void)hipHostFree(t_dl.pinned);
    t_dl.pinned = nullptr;
    t_dl.cap = 0;
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, cap, hipHostMallocDefault);
    if (e != hipSuccess) return hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s);
    t_dl.pinned = static_cast<char*>(q);
    t_dl.cap = cap;
  }
  hipError_t e = hipMemcpyAsync(t_dl.pinned + t_dl.used, dev_src, bytes, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  t_dl.items.push_back({host_dst, t_dl.used, bytes});
  t_dl.used = need;
  return hipSuccess;
}

hipError_t download_sync(hipStream_t s) {
  hipError_t e = hipStreamSynchronize(s);
  for (const auto& it : t_dl.items) memcpy(it.dst, t_dl.pinned + it.off, it.n);
  t_dl.items.clear();
  t_dl.used = 0;
  return e;
}

// ---- profiling -----------------------------------------------------------------------------
static const char* kProfNames[] = {"conv",    "ransac_eval", "ransac_hyp", "knn",       "chamfer",
                                   "topk",    "symcut",      "kmap",       "ransac_pre"};
static constexpr int kNumProf = sizeof(kProfNames) / sizeof(kProfNames[0]);
static int g_prof_on = 0;
static std::atomic<uint64_t> g_prof_epoch{1};   // bumped by cs_prof_reset: deferred units of an older region are dropped
static std::mutex g_prof_mu;  // calls may come from several host threads (one stream each)
struct ProfPending {
  hipEvent_t e0, e1;
};
static std::vector<ProfPending> g_pending[kNumProf];
static std::vector<hipEvent_t> g_event_pool;   // timing events of finished scopes, reused (under g_prof_mu)
static double g_prof_ms[kNumProf] = {0};
static int64_t g_prof_n[kNumProf] = {0};
static double g_prof_units[kNumProf] = {0};

static int prof_id(const char* name) {
  for (int i = 0; i < kNumProf; ++i)
    if (strcmp(name, kProfNames[i]) == 0) return i;
  return -1;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double units) : id(-1), stream(s) {
  if (!g_prof_on) return;
  id = prof_id(name);
  if (id < 0) return;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_units[id] += units;
  }
  // events come from a free list (refilled by prof_drain): two hipEventCreate per scope were ~1 ms of host time per chair step
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_event_pool.size() >= 2) {
      e0 = g_event_pool.back();
      g_event_pool.pop_back();
      e1 = g_event_pool.back();
      g_event_pool.pop_back();
    }
  }
  if (!e0 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) {
    id = -1;
    return;
  }
  (void)hipEventRecord(e0, stream);
}

bool prof_enabled() { return g_prof_on != 0; }

void prof_add_units(const char* name, double units) {
  if (!g_prof_on) return;
  const int id = prof_id(name);
  if (id < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_units[id] += units;
}

uint64_t prof_epoch() { return g_prof_epoch.load(); }

void prof_add_units_epoch(const char* name, double units, uint64_t epoch) {
  if (epoch != g_prof_epoch.load()) return;   // the region these units belong to has been reset away
  const int id = prof_id(name);
  if (id < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_units[id] += units;
}

hipStream_t side_stream(int which) {
  constexpr int N = 4;
  thread_local hipStream_t st[N] = {nullptr, nullptr, nullptr, nullptr};
  thread_local int st_dev = -1;
  if (which < 0 || which >= N) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (dev != st_dev) {  // first use (or the thread moved to another device: the old streams are abandoned)
    st_dev = dev;
    int lo = 0, hi = 0;  // numerically greatest = lowest priority
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
    for (int i = 0; i < N; ++i) {
      st[i] = nullptr;
      if (hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, lo) != hipSuccess) st[i] = nullptr;
    }
  }
  return st[which];
}

ProfScope::~ProfScope() {
  if (id < 0) return;
  (void)hipEventRecord(e1, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_pending[id].push_back({e0, e1});
}

static void prof_drain(int id) {
  std::vector<ProfPending> pend;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    pend.swap(g_pending[id]);
  }
  for (auto& p : pend) {
    if (hipEventSynchronize(p.e1) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
        g_prof_ms[id] += ms;
        g_prof_n[id] += 1;
      }
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_event_pool.size() < 16384) {
      g_event_pool.push_back(p.e0);
      g_event_pool.push_back(p.e1);
    } else {
      (void)hipEventDestroy(p.e0);
      (void)hipEventDestroy(p.e1);
    }
  }
}

}  // namespace cs

extern "C" {

const char* cs_last_error(void) { return cs::g_err; }

int cs_version(void) { return 100; }

int cs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void cs_pool_trim(void) { cs::trim_thread_cache(); }

void cs_pool_stats(uint64_t out[3]) {
  unsigned long long v[3];
  cs::pool_stats(v);
  for (int i = 0; i < 3; ++i) out[i] = v[i];
}

void cs_prof_enable(int on) {
  if (on) {
    // the events of the region ahead are created here, outside of what is being measured (a scope takes two from the pool)
    std::lock_guard<std::mutex> lk(cs::g_prof_mu);
    while (cs::g_event_pool.size() < 8192) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) break;
      cs::g_event_pool.push_back(e);
    }
  }
  // closing a region: maps whose pair count was still on its way when their convolutions were bracketed deliver those
  // units NOW -- not when Python happens to free the map, possibly inside the next leg's region
  if (!on) cs::kernelmap_flush_prof();
  cs::g_prof_on = on;
}

void cs_prof_reset(void) {
  cs::g_prof_epoch.fetch_add(1);
  for (int i = 0; i < cs::kNumProf; ++i) {
    cs::prof_drain(i);
    cs::g_prof_ms[i] = 0;
    cs::g_prof_n[i] = 0;
    cs::g_prof_units[i] = 0;
  }
}

int cs_prof_get(const char* name, double* total_ms, int64_t* launches) {
  int id = cs::prof_id(name);
  if (id < 0) {
    cs::set_error("unknown profile family '%s'", name);
    return CS_ERR_INVALID;
  }
  cs::prof_drain(id);
  if (total_ms) *total_ms = cs::g_prof_ms[id];
  if (launches) *launches = cs::g_prof_n[id];
  return CS_OK;
}

int cs_prof_get_units(const char* name, double* units) {
  int id = cs::prof_id(name);
  if (id < 0) {
    cs::set_error("unknown profile family '%s'", name);
    return CS_ERR_INVALID;
  }
  cs::kernelmap_flush_prof();
  if (units) *units = cs::g_prof_units[id];
  return CS_OK;
}

}  // extern "C"
