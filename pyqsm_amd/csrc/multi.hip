// multi.hip — the ray sweep over several MI355X: rays sharded contiguously, mesh replicated,
// results all-gathered, with RCCL (over xGMI inside a node) behind the C-ABI. Stands in for
// scene.cast_rays(rays) at pyQSM/viz/ray_casting.py:275-279 when more than one GPU is used
// (SURVEY.md §8b "Multi-GPU only inside pyqsm_cast_rays", §8e).
//
// Two ways to own the GPUs:
//   * one PROCESS for all of them: pyqsm_cast_rays_multi (ncclCommInitAll, one host thread per
//     device for the duration of the call). Host buffers in, host buffers out.
//   * one process PER GPU (how bench.py is launched): pyqsm_comm_unique_id /
//     pyqsm_comm_init_rank build one communicator per process; pyqsm_comm_broadcast_dev /
//     pyqsm_comm_all_gather_dev / pyqsm_comm_all_reduce_max are the data-path collectives on
//     the library stream, device pointers in and out.
// The only exchanges are the broadcast of the expanded mesh (48 B per triangle: 24 MB for
// 500 k) and the all-gather of 8 (16 with uv) bytes per ray; the sweep itself needs none.
// xGMI is point-to-point: a ring all-gather of 80 MB over 8 GPUs moves 70 MB per link at
// ~50-100 GB/s effective = ~1 ms against ~100 ms of brute-force sweep per shard.
//
// LOGICAL RANKS (PYQSM_MULTI_FAKE_RANKS=N, a test mode): RCCL refuses two ranks on one device,
// so on a one-GPU box the orchestration above — per-rank host thread, stream, arena, shard,
// rendezvous, ragged result blocks, failure hand-shake — would never run with more than one
// rank before the first 8-GPU node sees it. With the variable set the box pretends to have N
// devices, all of them device 0 (rank r -> device r % real count): every rank keeps its own
// host thread, context (stream + arena) and shard, and ONLY the three collectives are replaced,
// behind the same call sites (Coll below), by a host barrier + device-to-device copies between
// the ranks' buffers. Both entry families honour it; for pyqsm_comm_* the "processes" are host
// threads of one process (a communicator per thread instead of per process).
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <thread>

#include "raycast.hpp"

namespace pyqsm {

#define PQ_NCCL(expr)                                                                     \
  do {                                                                                    \
    ncclResult_t r__ = (expr);                                                            \
    if (r__ != ncclSuccess)                                                               \
      return ::pyqsm::fail(PYQSM_EHIP, "%s failed: %s (%s:%d)", #expr,                    \
                           ncclGetErrorString(r__), __FILE__, __LINE__);                  \
  } while (0)

// [begin, end) of rank r's contiguous shard; sizes differ by at most one (parallel.py: shard_bounds)
static void shard(int64_t n, int world, int rank, int64_t* b, int64_t* e) {
  const int64_t base = n / world, extra = n % world;
  *b = rank * base + std::min<int64_t>(rank, extra);
  *e = *b + base + (rank < extra ? 1 : 0);
}

// PYQSM_MULTI_FAKE_RANKS: 0 = off (the default), N >= 1 = N logical devices
static int fake_ranks() {
  const char* s = getenv("PYQSM_MULTI_FAKE_RANKS");
  if (!s || !*s) return 0;
  const long v = strtol(s, nullptr, 10);
  return v < 0 ? 0 : int(std::min<long>(v, 64));
}

// counting barrier for host threads (the per-device threads of one call; the logical ranks)
class Rendezvous {
 public:
  explicit Rendezvous(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    const int gen = gen_;
    if (++arrived_ == n_) {
      arrived_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return gen_ != gen; });
    }
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, arrived_ = 0, gen_ = 0;
};

// ---- the collectives, real or between logical ranks ----------------------------------------
struct FakeWorld {
  explicit FakeWorld(int n_) : n(n_), meet(n_), ptr(size_t(n_), nullptr), val(size_t(n_), 0.0) {}
  int n;
  Rendezvous meet;
  std::vector<const void*> ptr;  // what each rank published for the collective in flight
  std::vector<double> val;
  std::atomic<int> err{0};  // sticky: a failed copy fails every later collective of this world
  int joined = 0, left = 0;  // pyqsm_comm_* bookkeeping (under g_comm_mu)
};

struct Coll {  // one rank's handle; exactly one of comm / fake is set
  ncclComm_t comm = nullptr;
  FakeWorld* fake = nullptr;
  int rank = 0, world = 1;
};

// Test hook for the failure hand-shake: PYQSM_MULTI_INJECT_FAIL="<rank>,<op>" makes that rank's
// broadcast (op 1) or all-gather (op 2) fail as if it could not be enqueued.
static bool injected_failure(const Coll& k, int op) {
  const char* s = getenv("PYQSM_MULTI_INJECT_FAIL");
  int r = -1, o = -1;
  return s && sscanf(s, "%d,%d", &r, &o) == 2 && r == k.rank && o == op;
}

static int fake_done(FakeWorld& w, hipError_t e, const char* what) {
  if (e != hipSuccess) w.err.store(1);
  w.meet.wait();  // every rank's copies are finished (or failed) before any buffer is reused
  if (w.err.load() != 0)
    return fail(PYQSM_EHIP, "%s between logical ranks failed%s%s", what, e != hipSuccess ? ": " : "",
                e != hipSuccess ? hipGetErrorString(e) : " on another rank");
  return 0;
}

static int coll_broadcast(const Coll& k, void* buf, size_t bytes, int root, hipStream_t s) {
  const bool inject = injected_failure(k, 1);
  if (!k.fake) {
    if (inject) return fail(PYQSM_EHIP, "ncclBroadcast: injected failure on rank %d", k.rank);
    PQ_NCCL(ncclBroadcast(buf, buf, bytes, ncclUint8, root, k.comm, s));
    return 0;
  }
  FakeWorld& w = *k.fake;
  hipError_t e = inject ? hipErrorUnknown : hipSuccess;
  w.ptr[size_t(k.rank)] = buf;
  if (k.rank == root && !inject) e = hipStreamSynchronize(s);  // the payload is complete before anyone copies it
  if (e != hipSuccess) w.err.store(1);
  w.meet.wait();
  if (k.rank != root && w.err.load() == 0) {
    e = hipMemcpyAsync(buf, w.ptr[size_t(root)], bytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  return fake_done(w, e, "broadcast");
}

static int coll_all_gather(const Coll& k, const void* send, void* recv, size_t bytes_per_rank,
                           hipStream_t s) {
  const bool inject = injected_failure(k, 2);
  if (!k.fake) {
    if (inject) return fail(PYQSM_EHIP, "ncclAllGather: injected failure on rank %d", k.rank);
    PQ_NCCL(ncclAllGather(send, recv, bytes_per_rank, ncclUint8, k.comm, s));
    return 0;
  }
  FakeWorld& w = *k.fake;
  w.ptr[size_t(k.rank)] = send;
  hipError_t e = inject ? hipErrorUnknown : hipStreamSynchronize(s);  // this rank's block is complete
  if (e != hipSuccess) w.err.store(1);
  w.meet.wait();
  if (w.err.load() == 0) {
    for (int q = 0; q < w.n && e == hipSuccess; ++q) {
      char* dst = static_cast<char*>(recv) + size_t(q) * bytes_per_rank;
      if (dst == w.ptr[size_t(q)]) continue;  // in place: the rank's own block
      e = hipMemcpyAsync(dst, w.ptr[size_t(q)], bytes_per_rank, hipMemcpyDeviceToDevice, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  return fake_done(w, e, "all-gather");
}

// ---- one process, all devices ------------------------------------------------------------
struct LocalComms {
  std::vector<ncclComm_t> comms;
};
static std::mutex g_comm_mu;
static std::map<int, LocalComms*>& local_sets() {
  static auto* m = new std::map<int, LocalComms*>();
  return *m;
}

static int local_comms(int n, LocalComms** out) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  auto it = local_sets().find(n);
  if (it != local_sets().end()) {
    *out = it->second;
    return 0;
  }
  auto* lc = new LocalComms();
  lc->comms.resize(size_t(n));
  std::vector<int> devs(static_cast<size_t>(n));
  for (int d = 0; d < n; ++d) devs[size_t(d)] = d;
  ncclResult_t r = ncclCommInitAll(lc->comms.data(), n, devs.data());
  if (r != ncclSuccess) {
    delete lc;
    return fail(PYQSM_EHIP, "ncclCommInitAll(%d devices) failed: %s", n, ncclGetErrorString(r));
  }
  local_sets()[n] = lc;
  *out = lc;
  return 0;
}

// A collective that could not be enqueued on one device leaves the others' kernels waiting for
// a peer that never comes: abort every communicator of the set (that unblocks them) and forget
// the set, so that the next call initialises a fresh one.
static void abort_local_set(int n, LocalComms* lc) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  auto it = local_sets().find(n);
  if (it == local_sets().end() || it->second != lc) return;  // another thread was first
  for (ncclComm_t c : lc->comms) (void)ncclCommAbort(c);
  local_sets().erase(it);
  delete lc;
}

struct DevJob {
  int rc = 0;
  std::string err;
};

// ---- one process per device -------------------------------------------------------------
struct RankComm {
  Coll k;
  int device = 0;
  std::string id;  // logical ranks: key of the world in fake_worlds()
  bool active() const { return k.comm != nullptr || k.fake != nullptr; }
};
static RankComm g_rank;               // RCCL: one communicator per process
static thread_local RankComm t_rank;  // logical ranks: one per host thread
static std::map<std::string, FakeWorld*>& fake_worlds() {
  static auto* m = new std::map<std::string, FakeWorld*>();
  return *m;
}
static std::condition_variable g_join_cv;  // logical ranks: init_rank is collective

static RankComm* current_rank() { return t_rank.active() ? &t_rank : &g_rank; }

static void leave_fake_world(RankComm* rc) {  // g_comm_mu held
  auto it = fake_worlds().find(rc->id);
  if (it != fake_worlds().end() && it->second == rc->k.fake && ++it->second->left == it->second->n) {
    delete it->second;
    fake_worlds().erase(it);
  }
  *rc = RankComm();
}

void comm_shutdown() {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  for (auto& kv : local_sets()) {
    for (ncclComm_t c : kv.second->comms) (void)ncclCommDestroy(c);
    delete kv.second;
  }
  local_sets().clear();
  if (g_rank.k.comm) {
    (void)ncclCommDestroy(g_rank.k.comm);
    g_rank = RankComm();
  }
  if (t_rank.active()) leave_fake_world(&t_rank);
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_cast_rays_multi(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                          const float* rays, int64_t R, float* t_hit, uint32_t* prim_id, float* uv,
                          int32_t n_devices) {
  PQ_API_RANGE("pyqsm_cast_rays_multi");
  PQ_TRY(ray_check_sizes(V, T, R));
  if (R == 0) return 0;
  if (!rays || !t_hit || !prim_id || (T > 0 && (!verts || !tris)))
    return fail(PYQSM_EINVAL, "pyqsm_cast_rays_multi: NULL pointer");
  const int real = pyqsm_device_count();
  if (real <= 0) return fail(PYQSM_ENODEV, "no HIP device available");
  const int logical = fake_ranks();
  const int avail = logical > 0 ? logical : real;
  if (n_devices <= 0) n_devices = avail;
  if (n_devices > avail)
    return fail(PYQSM_ENODEV, "%d devices asked for, %d visible", int(n_devices), avail);
  const int n = int(std::min<int64_t>(n_devices, R));  // never more ranks than rays
  LocalComms* lc = nullptr;
  FakeWorld fake(n);
  if (logical == 0) PQ_TRY(local_comms(n, &lc));
  const int W = uv ? 4 : 2;  // 32-bit words per ray in a result block: t, prim [, u, v]
  const int64_t cap = (R + n - 1) / n;
  std::vector<DevJob> jobs(static_cast<size_t>(n));
  Rendezvous meet(n);
  std::atomic<int> failed{0}, coll_failed{0};
  auto work = [&](int d) {
    DevJob& job = jobs[size_t(d)];
    auto run = [&]() -> int {
      Ctx* c = ctx_for(logical > 0 ? d % real : d);
      if (!c) {
        failed.fetch_add(1);
        meet.wait();
        return PYQSM_ENODEV;
      }
      std::lock_guard<std::mutex> lk(c->mu);
      int64_t b, e;
      shard(R, n, d, &b, &e);
      const int64_t r_loc = e - b;
      float *tri12 = nullptr, *d_rays = nullptr;
      uint32_t* block = nullptr;
      // ---- phase 1: everything that can fail for local reasons -----------------------
      auto phase1 = [&]() -> int {
        c->arena.reset();
        PQ_TRY(c->arena.get(size_t(T) * 12 + 4, &tri12));
        PQ_TRY(c->arena.get(size_t(r_loc) * 6 + 6, &d_rays));
        PQ_TRY(c->arena.get(size_t(n) * W * cap + 4, &block));
        if (r_loc)
          PQ_HIP(hipMemcpyAsync(d_rays, rays + 6 * b, size_t(r_loc) * 24, hipMemcpyHostToDevice,
                                c->stream));
        if (d == 0 && T > 0) {
          float* d_verts;
          int32_t* d_tris;
          PQ_TRY(c->arena.get(size_t(V) * 3 + 1, &d_verts));
          PQ_TRY(c->arena.get(size_t(T) * 3 + 1, &d_tris));
          PQ_HIP(hipMemcpyAsync(d_verts, verts, size_t(V) * 12, hipMemcpyHostToDevice, c->stream));
          PQ_HIP(hipMemcpyAsync(d_tris, tris, size_t(T) * 12, hipMemcpyHostToDevice, c->stream));
          PQ_TRY(ray_expand(c, d_verts, V, d_tris, T, tri12));  // checks the indices, synchronises
        }
        return 0;
      };
      int rc = phase1();
      if (rc != 0) failed.fetch_add(1);
      meet.wait();  // nobody enters a collective unless everybody can
      if (failed.load() != 0) return rc;
      // ---- phase 2: broadcast, sweep, all-gather. From here on a thread leaves only where
      // every thread leaves (all_enqueued): the others wait for it in every collective ---------
      Coll k;
      k.rank = d;
      k.world = n;
      if (logical > 0)
        k.fake = &fake;
      else
        k.comm = lc->comms[size_t(d)];
      // A collective is only ENQUEUED by its call; its kernel then waits for the peers' kernels.
      // If the call fails on one device, the others must not synchronise their streams (the
      // sweep's setup does) before they know: after each collective the threads meet on the
      // host, and if anybody's call failed all of them abort the communicator set — that ends
      // the waiting kernels — drain their streams and return the error.
      auto all_enqueued = [&](int rc_coll) -> int {
        const std::string msg = rc_coll != 0 ? pyqsm_last_error() : "";
        if (rc_coll != 0) coll_failed.fetch_add(1);
        meet.wait();
        if (coll_failed.load() == 0) return 0;
        if (lc) abort_local_set(n, lc);
        (void)hipStreamSynchronize(c->stream);
        return rc_coll != 0 ? fail(rc_coll, "%s", msg.c_str())
                            : fail(PYQSM_EHIP, "a collective failed on another device");
      };
      if (T > 0 && n > 1) PQ_TRY(all_enqueued(coll_broadcast(k, tri12, size_t(T) * 48, 0, c->stream)));
      uint32_t* mine = block + size_t(d) * W * cap;
      rc = ray_launch(c, tri12, T, d_rays, r_loc, reinterpret_cast<float*>(mine), mine + cap,
                      uv ? reinterpret_cast<float*>(mine + 2 * cap) : nullptr);
      // a failed launch still takes part in the all-gather: the other devices are waiting in it
      PQ_TRY(all_enqueued(coll_all_gather(k, mine, block, size_t(W) * cap * 4, c->stream)));
      if (rc != 0) return rc;
      if (d == 0) {
        for (int r = 0; r < n; ++r) {
          int64_t rb, re;
          shard(R, n, r, &rb, &re);
          if (re == rb) continue;
          const uint32_t* src = block + size_t(r) * W * cap;
          PQ_HIP(hipMemcpyAsync(t_hit + rb, src, size_t(re - rb) * 4, hipMemcpyDeviceToHost, c->stream));
          PQ_HIP(hipMemcpyAsync(prim_id + rb, src + cap, size_t(re - rb) * 4, hipMemcpyDeviceToHost,
                                c->stream));
          if (uv)
            PQ_HIP(hipMemcpyAsync(uv + 2 * rb, src + 2 * cap, size_t(re - rb) * 8,
                                  hipMemcpyDeviceToHost, c->stream));
        }
      }
      PQ_HIP(hipStreamSynchronize(c->stream));
      return 0;
    };
    job.rc = run();
    if (job.rc != 0) job.err = pyqsm_last_error();
  };
  std::vector<std::thread> threads;
  for (int d = 1; d < n; ++d) threads.emplace_back(work, d);
  work(0);
  for (auto& t : threads) t.join();
  for (int d = 0; d < n; ++d)
    if (jobs[size_t(d)].rc != 0)
      return fail(jobs[size_t(d)].rc, "device %d: %s", d, jobs[size_t(d)].err.c_str());
  return 0;
}

int pyqsm_shard_bounds(int64_t n, int32_t world, int32_t rank, int64_t* begin, int64_t* end) {
  if (n < 0 || world < 1 || rank < 0 || rank >= world || !begin || !end)
    return fail(PYQSM_EINVAL, "pyqsm_shard_bounds: bad argument");
  shard(n, world, rank, begin, end);
  return 0;
}

/* ---- one process per GPU ------------------------------------------------------------- */

int pyqsm_comm_unique_id(uint8_t* id) {
  if (!id) return fail(PYQSM_EINVAL, "pyqsm_comm_unique_id: NULL pointer");
  if (fake_ranks() > 0) {  // logical ranks: any process-unique 128 bytes do
    static std::atomic<uint64_t> serial{0};
    memset(id, 0, PYQSM_COMM_ID_BYTES);
    const uint64_t words[2] = {serial.fetch_add(1) + 1,
                               uint64_t(std::chrono::steady_clock::now().time_since_epoch().count())};
    memcpy(id, "pyqsm-logical-ranks", 19);
    memcpy(id + 32, words, sizeof(words));
    return 0;
  }
  ncclUniqueId u;
  PQ_NCCL(ncclGetUniqueId(&u));
  static_assert(sizeof(u) == PYQSM_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id, &u, sizeof(u));
  return 0;
}

int pyqsm_comm_init_rank(const uint8_t* id, int32_t world, int32_t rank, int32_t device) {
  if (!id) return fail(PYQSM_EINVAL, "pyqsm_comm_init_rank: NULL pointer");
  if (world < 1 || rank < 0 || rank >= world) return fail(PYQSM_EINVAL, "bad world / rank");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  const int logical = fake_ranks();
  if (logical > 0) {
    if (world > logical)
      return fail(PYQSM_EINVAL, "world %d exceeds PYQSM_MULTI_FAKE_RANKS=%d", int(world), logical);
    std::unique_lock<std::mutex> lk(g_comm_mu);
    if (t_rank.active()) return fail(PYQSM_EINVAL, "a communicator already exists in this thread");
    const std::string key(reinterpret_cast<const char*>(id), PYQSM_COMM_ID_BYTES);
    FakeWorld*& w = fake_worlds()[key];
    if (!w) w = new FakeWorld(world);
    FakeWorld* mine = w;
    if (mine->n != world || mine->joined >= mine->n) return fail(PYQSM_EINVAL, "world size mismatch for this id");
    ++mine->joined;
    t_rank.k.fake = mine;
    t_rank.k.rank = rank;
    t_rank.k.world = world;
    t_rank.device = device;
    t_rank.id = key;
    g_join_cv.notify_all();
    // collective, like ncclCommInitRank: returns once every rank has joined (bounded wait)
    if (!g_join_cv.wait_for(lk, std::chrono::seconds(120), [&] { return mine->joined == mine->n; })) {
      leave_fake_world(&t_rank);
      return fail(PYQSM_EHIP, "logical ranks: not every rank joined within 120 s");
    }
    return 0;
  }
  std::lock_guard<std::mutex> lk(g_comm_mu);
  if (g_rank.k.comm) return fail(PYQSM_EINVAL, "a communicator already exists in this process");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  PQ_NCCL(ncclCommInitRank(&g_rank.k.comm, world, u, rank));
  g_rank.k.world = world;
  g_rank.k.rank = rank;
  g_rank.device = device;
  return 0;
}

int pyqsm_comm_finalize(void) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  if (t_rank.active()) {
    leave_fake_world(&t_rank);
    return 0;
  }
  if (g_rank.k.comm) {
    (void)hipSetDevice(g_rank.device);
    (void)ncclCommDestroy(g_rank.k.comm);
    g_rank = RankComm();
  }
  return 0;
}

int pyqsm_comm_info(int32_t* world, int32_t* rank, int32_t* device) {
  const RankComm* rc = current_rank();
  if (world) *world = rc->active() ? rc->k.world : 0;
  if (rank) *rank = rc->k.rank;
  if (device) *device = rc->device;
  return 0;
}

static int rank_ctx(RankComm** rc, Ctx** c) {
  *rc = current_rank();
  if (!(*rc)->active()) return fail(PYQSM_EINVAL, "no communicator: call pyqsm_comm_init_rank first");
  *c = ctx_for((*rc)->device);
  return *c ? 0 : PYQSM_ENODEV;
}

int pyqsm_comm_broadcast_dev(void* buf_dev, int64_t bytes, int32_t root) {
  PQ_API_RANGE("pyqsm_comm_broadcast_dev");
  RankComm* rc;
  Ctx* c;
  PQ_TRY(rank_ctx(&rc, &c));
  if (bytes < 0 || root < 0 || root >= rc->k.world) return fail(PYQSM_EINVAL, "bad size / root");
  if (bytes == 0) return 0;
  if (!buf_dev) return fail(PYQSM_EINVAL, "pyqsm_comm_broadcast_dev: NULL pointer");
  std::lock_guard<std::mutex> lk(c->mu);
  return coll_broadcast(rc->k, buf_dev, size_t(bytes), root, c->stream);
}

int pyqsm_comm_all_gather_dev(const void* send_dev, void* recv_dev, int64_t bytes_per_rank) {
  PQ_API_RANGE("pyqsm_comm_all_gather_dev");
  RankComm* rc;
  Ctx* c;
  PQ_TRY(rank_ctx(&rc, &c));
  if (bytes_per_rank < 0) return fail(PYQSM_EINVAL, "negative size");
  if (bytes_per_rank == 0) return 0;
  if (!send_dev || !recv_dev) return fail(PYQSM_EINVAL, "pyqsm_comm_all_gather_dev: NULL pointer");
  std::lock_guard<std::mutex> lk(c->mu);
  return coll_all_gather(rc->k, send_dev, recv_dev, size_t(bytes_per_rank), c->stream);
}

int pyqsm_comm_all_reduce_max(double* value) {
  PQ_API_RANGE("pyqsm_comm_all_reduce_max");
  RankComm* rc;
  Ctx* c;
  PQ_TRY(rank_ctx(&rc, &c));
  if (!value) return fail(PYQSM_EINVAL, "pyqsm_comm_all_reduce_max: NULL pointer");
  std::lock_guard<std::mutex> lk(c->mu);
  if (rc->k.fake) {
    FakeWorld& w = *rc->k.fake;
    hipError_t e = hipStreamSynchronize(c->stream);  // what the RCCL kernel's place on the stream gives
    w.val[size_t(rc->k.rank)] = *value;
    w.meet.wait();
    double m = w.val[0];
    for (int q = 1; q < w.n; ++q) m = std::max(m, w.val[size_t(q)]);
    PQ_TRY(fake_done(w, e, "all-reduce"));
    *value = m;
    return 0;
  }
  double* d;
  PQ_HIP(hipMalloc(&d, 8));  // not from the arena: other calls' scratch may be live there
  hipError_t e = hipMemcpyAsync(d, value, 8, hipMemcpyHostToDevice, c->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = ncclAllReduce(d, d, 1, ncclDouble, ncclMax, rc->k.comm, c->stream);
  if (e == hipSuccess && r == ncclSuccess)
    e = hipMemcpyAsync(value, d, 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (r != ncclSuccess) return fail(PYQSM_EHIP, "ncclAllReduce failed: %s", ncclGetErrorString(r));
  if (e != hipSuccess) return fail(PYQSM_EHIP, "all_reduce_max: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"
