// One process, several GPUs: a single proof sharded over the devices of a node (BASELINE config 4) for
// hosts that keep ONE process -- the Node.js addon of BASELINE.json's north_star.  (bench.py, one process per GPU,
// drives the same pipeline through g16_shard_begin / g16_shard_end with an RCCL scatter in the middle.)
//
// A g16_multi owns one sharded g16_prover per device and one host thread per shard, alive for the life of the handle
// (r02 created 2 G threads per proof).  g16_multi_prove is three fork-joins of those threads; between them only
// EVENTS order the devices -- no blocking copy, no drained stream:
//   1. shard 0 uploads the witness from the caller's buffer (the only host-to-device copy of a proof: r02 had every
//      shard upload its own 26 MB from pageable memory) and records an event;
//   2. every other shard makes its main stream wait for that event and pulls the witness from shard 0 with a peer copy
//      (xGMI between the GPUs of a node; a plain device copy when two shards share a GPU); every shard then enqueues its
//      witness MSMs, and shard v mod G the odd-coset evaluation of vector v of (A, B, C), recording an event per vector;
//   3. every shard makes its main stream wait for the three vector events, pulls its slice [lo_r, hi_r) of each vector
//      straight into its own vector buffers (asynchronous peer copies on ITS stream: the 3 G copies of a proof run side
//      by side on G streams instead of one after another on the host thread), joins, runs the H-MSM over its bases and
//      folds its partial sums;
//   4. the 768-byte partial blobs are added and the proof finished on the host (g16_prove_finish).
// Nothing is replicated except the QAP evaluation on the (at most three) shards that own a vector.
#include <hip/hip_runtime.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g16_prover.h"
#include "internal.h"

using namespace g16;

namespace {

// G host threads, one per shard, parked on a condition variable between jobs
class ShardPool {
 public:
  explicit ShardPool(size_t n) : rc_(n, 0), err_(n) {
    for (size_t k = 0; k < n; k++) th_.emplace_back([this, k] { loop(k); });
  }
  ~ShardPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // fn(k) on every shard's thread; returns when all are done.  The first failure (code + that thread's error text) wins.
  int run(const std::function<int(size_t)>& fn) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      pending_ = th_.size();
      gen_++;
    }
    cv_.notify_all();
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_.wait(lk, [this] { return pending_ == 0; });
      fn_ = nullptr;
    }
    for (size_t k = 0; k < rc_.size(); k++)
      if (rc_[k]) {
        set_error("shard " + std::to_string(k) + ": " + err_[k]);
        return rc_[k];
      }
    return G16_OK;
  }

 private:
  void loop(size_t k) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<int(size_t)>* fn;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_;
      }
      rc_[k] = (*fn)(k);
      err_[k] = rc_[k] ? g16_last_error() : "";
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::vector<std::thread> th_;
  std::vector<int> rc_;
  std::vector<std::string> err_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<int(size_t)>* fn_ = nullptr;
  size_t pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

}  // namespace

struct g16_multi {
  std::vector<g16_prover*> h;
  std::vector<int> dev;
  std::vector<ShardView> view;                 // per shard: its stream, witness slot and vector buffers
  hipEvent_t ev_w = nullptr;                   // witness resident on shard 0
  hipEvent_t ev_vec[3] = {nullptr, nullptr, nullptr};   // vector v evaluated on shard v mod G
  uint32_t N = 0, n_public = 0;
  std::unique_ptr<ShardPool> pool;
  std::mutex mu;

  ~g16_multi() {
    pool.reset();   // (joins the threads before the handles they use go away)
    if (!dev.empty()) {
      (void)hipSetDevice(dev[0]);
      if (ev_w) (void)hipEventDestroy(ev_w);
      for (uint32_t v = 0; v < 3; v++)
        if (ev_vec[v]) {
          (void)hipSetDevice(dev[v % dev.size()]);
          (void)hipEventDestroy(ev_vec[v]);
        }
    }
    for (auto* p : h)
      if (p) g16_destroy(p);
  }
};

// dst (on shard `to`'s device) <- src (on shard `from`'s device), enqueued on shard `to`'s main stream
static int shard_copy(const g16_multi* m, size_t to, void* dst, size_t from, const void* src, size_t bytes) {
  if (!bytes) return G16_OK;
  hipError_t e = m->dev[to] == m->dev[from]
                     ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, m->view[to].st)
                     : hipMemcpyPeerAsync(dst, m->dev[to], src, m->dev[from], bytes, m->view[to].st);
  if (e != hipSuccess) { set_error(std::string("peer copy failed: ") + hipGetErrorString(e)); return G16_E_HIP; }
  return G16_OK;
}

extern "C" {

int g16_multi_create(const uint8_t* zkey, size_t zkey_len, const int32_t* devices, uint32_t ndev, const g16_opts* opts,
                     g16_multi** out) {
  if (!out) { set_error("out is NULL"); return G16_E_ARG; }
  *out = nullptr;
  if (!zkey || !devices || ndev == 0 || ndev > 64) { set_error("g16_multi_create: 1..64 devices"); return G16_E_ARG; }
  std::unique_ptr<g16_multi> M(new g16_multi());
  M->h.assign(ndev, nullptr);
  M->dev.assign(devices, devices + ndev);
  M->view.assign(ndev, ShardView());
  M->pool.reset(new ShardPool(ndev));
  int rc = M->pool->run([&](size_t k) {
    g16_opts o{};
    if (opts) o = *opts;
    o.device = M->dev[k];
    o.shard_rank = (int32_t)k;
    o.shard_count = (int32_t)ndev;
    int e = g16_create(zkey, zkey_len, &o, &M->h[k]);
    if (e) return e;
    if ((e = shard_view(M->h[k], 0, &M->view[k]))) return e;
    // direct xGMI copies where the platform offers them (a refusal only means the copies are staged)
    for (uint32_t j = 0; j < ndev; j++)
      if (M->dev[j] != M->dev[k]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, M->dev[k], M->dev[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(M->dev[j], 0);
      }
    (void)hipGetLastError();
    return G16_OK;
  });
  if (rc) return rc;
  g16_info inf;
  if ((rc = g16_get_info(M->h[0], &inf))) return rc;
  M->N = inf.domain_size;
  M->n_public = inf.n_public;
  G16_HIP(hipSetDevice(M->dev[0]));
  G16_HIP(hipEventCreateWithFlags(&M->ev_w, hipEventDisableTiming));
  for (uint32_t v = 0; v < 3; v++) {
    G16_HIP(hipSetDevice(M->dev[v % ndev]));
    G16_HIP(hipEventCreateWithFlags(&M->ev_vec[v], hipEventDisableTiming));
  }
  *out = M.release();
  return G16_OK;
}

void g16_multi_destroy(g16_multi* m) { delete m; }

int g16_multi_prove(g16_multi* m, const uint8_t* wtns, size_t wtns_len, const uint8_t r[32], const uint8_t s[32],
                    g16_proof* out, uint8_t* pub) {
  if (!m || !wtns || !out) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(m->mu);
  const size_t G = m->h.size();
  // 1. the witness goes to shard 0 (format checks and the canonicity check of its words happen there)
  int rc = shard_upload_witness(m->h[0], 0, wtns, wtns_len);
  if (rc) return rc;
  if (hipSetDevice(m->dev[0]) != hipSuccess || hipEventRecord(m->ev_w, m->view[0].st) != hipSuccess) {
    set_error("hipEventRecord failed");
    return G16_E_HIP;
  }
  // 2. fan the witness out, start the witness MSMs and the vector evaluations
  rc = m->pool->run([&](size_t k) {
    if (hipSetDevice(m->dev[k]) != hipSuccess) { set_error("hipSetDevice failed"); return (int)G16_E_HIP; }
    int e;
    if (k != 0) {
      if (hipStreamWaitEvent(m->view[k].st, m->ev_w, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return (int)G16_E_HIP; }
      if ((e = shard_copy(m, k, m->view[k].d_w, 0, m->view[0].d_w, (size_t)m->view[k].nVars * 32))) return e;
    }
    uint32_t mask = 0;
    for (uint32_t v = 0; v < 3; v++)
      if (v % G == k) mask |= 1u << v;
    if ((e = shard_begin_async(m->h[k], 0, mask))) return e;
    // (one event per owned vector, all after the shard's whole coset evaluation: its vectors are transformed together)
    for (uint32_t v = 0; v < 3; v++)
      if ((mask >> v) & 1u)
        if (hipEventRecord(m->ev_vec[v], m->view[k].st) != hipSuccess) { set_error("hipEventRecord failed"); return (int)G16_E_HIP; }
    return (int)G16_OK;
  });
  // 3. slices to their shards, join + H-MSM + fold (after a failure above: drain what was begun)
  std::vector<uint8_t> parts(G * G16_PARTIAL_BYTES);
  if (rc) {
    const std::string first_err = g16_last_error();
    (void)m->pool->run([&](size_t k) { shard_drain(m->h[k]); return (int)G16_OK; });
    set_error(first_err);
    return rc;
  }
  rc = m->pool->run([&](size_t k) {
    if (hipSetDevice(m->dev[k]) != hipSuccess) { set_error("hipSetDevice failed"); return (int)G16_E_HIP; }
    const ShardView& me = m->view[k];
    int e = G16_OK;
    for (uint32_t v = 0; v < 3 && !e; v++) {
      const size_t owner = v % G;
      if (owner == k) continue;   // the slice is where the evaluation left it
      if (hipStreamWaitEvent(me.st, m->ev_vec[v], 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); e = G16_E_HIP; break; }
      e = shard_copy(m, k, me.vec[v] + me.lo, owner, m->view[owner].vec[v] + me.lo, (size_t)(me.hi - me.lo) * sizeof(F29));
    }
    if (e) { shard_drain(m->h[k]); return e; }
    return shard_end_collect(m->h[k], parts.data() + k * G16_PARTIAL_BYTES);
  });
  if (rc) return rc;
  if ((rc = shard_witness_verdict(m->h[0]))) return rc;
  // 4. add the partial sums, finish on the host
  return g16_prove_finish(m->h[0], 0, parts.data(), (uint32_t)G, r, s, out, pub);
}

int g16_multi_get_info(const g16_multi* m, g16_info* out, uint32_t* n_shards) {
  if (!m || !out) { set_error("NULL argument"); return G16_E_ARG; }
  if (n_shards) *n_shards = (uint32_t)m->h.size();
  return g16_get_info(m->h[0], out);
}

}  // extern "C"
