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
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
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

// counting barrier for the per-device host threads of one call
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

struct DevJob {
  int rc = 0;
  std::string err;
};

// ---- one process per device -------------------------------------------------------------
struct RankComm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
};
static RankComm g_rank;

void comm_shutdown() {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  for (auto& kv : local_sets()) {
    for (ncclComm_t c : kv.second->comms) (void)ncclCommDestroy(c);
    delete kv.second;
  }
  local_sets().clear();
  if (g_rank.comm) {
    (void)ncclCommDestroy(g_rank.comm);
    g_rank = RankComm();
  }
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
  const int avail = pyqsm_device_count();
  if (avail <= 0) return fail(PYQSM_ENODEV, "no HIP device available");
  if (n_devices <= 0) n_devices = avail;
  if (n_devices > avail)
    return fail(PYQSM_ENODEV, "%d devices asked for, %d visible", int(n_devices), avail);
  const int n = int(std::min<int64_t>(n_devices, R));  // never more ranks than rays
  LocalComms* lc = nullptr;
  PQ_TRY(local_comms(n, &lc));
  const int W = uv ? 4 : 2;  // 32-bit words per ray in a result block: t, prim [, u, v]
  const int64_t cap = (R + n - 1) / n;
  std::vector<DevJob> jobs(static_cast<size_t>(n));
  Rendezvous meet(n);
  std::atomic<int> failed{0};
  auto work = [&](int d) {
    DevJob& job = jobs[size_t(d)];
    auto run = [&]() -> int {
      Ctx* c = ctx_for(d);
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
      // ---- phase 2: broadcast, sweep, all-gather -----------------------------------------
      ncclComm_t comm = lc->comms[size_t(d)];
      if (T > 0 && n > 1)
        PQ_NCCL(ncclBroadcast(tri12, tri12, size_t(T) * 12, ncclFloat, 0, comm, c->stream));
      uint32_t* mine = block + size_t(d) * W * cap;
      rc = ray_launch(c, tri12, T, d_rays, r_loc, reinterpret_cast<float*>(mine), mine + cap,
                      uv ? reinterpret_cast<float*>(mine + 2 * cap) : nullptr);
      // a failed launch still takes part in the all-gather: the other devices are waiting in it
      PQ_NCCL(ncclAllGather(mine, block, size_t(W) * cap, ncclUint32, comm, c->stream));
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
  std::lock_guard<std::mutex> lk(g_comm_mu);
  if (g_rank.comm) return fail(PYQSM_EINVAL, "a communicator already exists in this process");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  PQ_NCCL(ncclCommInitRank(&g_rank.comm, world, u, rank));
  g_rank.world = world;
  g_rank.rank = rank;
  g_rank.device = device;
  return 0;
}

int pyqsm_comm_finalize(void) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  if (g_rank.comm) {
    (void)hipSetDevice(g_rank.device);
    (void)ncclCommDestroy(g_rank.comm);
    g_rank = RankComm();
  }
  return 0;
}

int pyqsm_comm_info(int32_t* world, int32_t* rank, int32_t* device) {
  if (world) *world = g_rank.comm ? g_rank.world : 0;
  if (rank) *rank = g_rank.rank;
  if (device) *device = g_rank.device;
  return 0;
}

static int rank_ctx(Ctx** c) {
  if (!g_rank.comm) return fail(PYQSM_EINVAL, "no communicator: call pyqsm_comm_init_rank first");
  *c = ctx_for(g_rank.device);
  return *c ? 0 : PYQSM_ENODEV;
}

int pyqsm_comm_broadcast_dev(void* buf_dev, int64_t bytes, int32_t root) {
  PQ_API_RANGE("pyqsm_comm_broadcast_dev");
  Ctx* c;
  PQ_TRY(rank_ctx(&c));
  if (bytes < 0 || root < 0 || root >= g_rank.world) return fail(PYQSM_EINVAL, "bad size / root");
  if (bytes == 0) return 0;
  if (!buf_dev) return fail(PYQSM_EINVAL, "pyqsm_comm_broadcast_dev: NULL pointer");
  std::lock_guard<std::mutex> lk(c->mu);
  PQ_NCCL(ncclBroadcast(buf_dev, buf_dev, size_t(bytes), ncclUint8, root, g_rank.comm, c->stream));
  return 0;
}

int pyqsm_comm_all_gather_dev(const void* send_dev, void* recv_dev, int64_t bytes_per_rank) {
  PQ_API_RANGE("pyqsm_comm_all_gather_dev");
  Ctx* c;
  PQ_TRY(rank_ctx(&c));
  if (bytes_per_rank < 0) return fail(PYQSM_EINVAL, "negative size");
  if (bytes_per_rank == 0) return 0;
  if (!send_dev || !recv_dev) return fail(PYQSM_EINVAL, "pyqsm_comm_all_gather_dev: NULL pointer");
  std::lock_guard<std::mutex> lk(c->mu);
  PQ_NCCL(ncclAllGather(send_dev, recv_dev, size_t(bytes_per_rank), ncclUint8, g_rank.comm, c->stream));
  return 0;
}

int pyqsm_comm_all_reduce_max(double* value) {
  PQ_API_RANGE("pyqsm_comm_all_reduce_max");
  Ctx* c;
  PQ_TRY(rank_ctx(&c));
  if (!value) return fail(PYQSM_EINVAL, "pyqsm_comm_all_reduce_max: NULL pointer");
  std::lock_guard<std::mutex> lk(c->mu);
  double* d;
  PQ_HIP(hipMalloc(&d, 8));  // not from the arena: other calls' scratch may be live there
  hipError_t e = hipMemcpyAsync(d, value, 8, hipMemcpyHostToDevice, c->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = ncclAllReduce(d, d, 1, ncclDouble, ncclMax, g_rank.comm, c->stream);
  if (e == hipSuccess && r == ncclSuccess)
    e = hipMemcpyAsync(value, d, 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (r != ncclSuccess) return fail(PYQSM_EHIP, "ncclAllReduce failed: %s", ncclGetErrorString(r));
  if (e != hipSuccess) return fail(PYQSM_EHIP, "all_reduce_max: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"
