// One process, several GPUs: a single proof sharded over the devices of a node (BASELINE config 4) for
// hosts that keep ONE process -- the Node.js addon of BASELINE.json's north_star.  (bench.py, one process per GPU,
// drives the same two-phase C ABI with an RCCL scatter in the middle.)
//
// A g16_multi owns one sharded g16_prover per device.  g16_multi_prove:
//   1. every shard, on its own host thread: stage the witness, g16_shard_begin -- the witness MSMs of its point
//      range start, and shard v mod G evaluates vector v of (A, B, C) on the odd coset into a buffer on its device;
//   2. the slices [lo_r, hi_r) of the three vectors go to shard r's device with peer copies (xGMI between the GPUs
//      of one node; a plain device copy when two shards share a GPU);
//   3. every shard: g16_shard_end -- join its slice of P, H-MSM over its H bases, partial sums back;
//   4. the 768-byte partial blobs are added and the proof finished on the host (g16_prove_finish).
// Nothing is replicated except the QAP evaluation on the (at most three) shards that own a vector.
#include <hip/hip_runtime.h>
#include <string.h>

#include <array>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g16_prover.h"
#include "internal.h"

using namespace g16;

struct g16_multi {
  std::vector<g16_prover*> h;
  std::vector<int> dev;
  uint32_t N = 0, n_public = 0;
  void* vec[3] = {nullptr, nullptr, nullptr};     // coset evaluations of A, B, C on the device of shard v mod G
  std::vector<std::array<void*, 3>> slice;        // per shard: its slices of the three vectors, on its device
  std::mutex mu;

  ~g16_multi() {
    for (size_t k = 0; k < h.size(); k++) {
      if (k < slice.size()) {
        (void)hipSetDevice(dev[k]);
        for (void* p : slice[k]) if (p) (void)hipFree(p);
      }
      if (h[k]) g16_destroy(h[k]);
    }
    for (int v = 0; v < 3; v++)
      if (vec[v] && !dev.empty()) {
        (void)hipSetDevice(dev[(size_t)v % dev.size()]);
        (void)hipFree(vec[v]);
      }
  }
};

// run fn(k) for every shard on its own host thread; the first failure (code + that thread's error text) wins
template <class Fn> static int for_each_shard(size_t count, Fn fn) {
  std::vector<int> rc(count, 0);
  std::vector<std::string> err(count);
  std::vector<std::thread> th;
  for (size_t k = 0; k < count; k++)
    th.emplace_back([&, k] {
      rc[k] = fn(k);
      if (rc[k]) err[k] = g16_last_error();
    });
  for (auto& t : th) t.join();
  for (size_t k = 0; k < count; k++)
    if (rc[k]) {
      set_error("shard " + std::to_string(k) + ": " + err[k]);
      return rc[k];
    }
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
  int rc = for_each_shard(ndev, [&](size_t k) {
    g16_opts o{};
    if (opts) o = *opts;
    o.device = M->dev[k];
    o.shard_rank = (int32_t)k;
    o.shard_count = (int32_t)ndev;
    return g16_create(zkey, zkey_len, &o, &M->h[k]);
  });
  if (rc) return rc;
  g16_info inf;
  if ((rc = g16_get_info(M->h[0], &inf))) return rc;
  M->N = inf.domain_size;
  M->n_public = inf.n_public;
  for (uint32_t v = 0; v < 3; v++) {
    G16_HIP(hipSetDevice(M->dev[v % ndev]));
    G16_HIP(hipMalloc(&M->vec[v], (size_t)M->N * G16_LAZY_FR_BYTES));
  }
  M->slice.assign(ndev, std::array<void*, 3>{nullptr, nullptr, nullptr});
  for (uint32_t k = 0; k < ndev; k++) {
    uint32_t lo, hi;
    g16_shard_range(M->N, (int32_t)k, (int32_t)ndev, &lo, &hi);
    G16_HIP(hipSetDevice(M->dev[k]));
    for (int v = 0; v < 3; v++) G16_HIP(hipMalloc(&M->slice[k][v], (size_t)(hi - lo + 1) * G16_LAZY_FR_BYTES));
    // direct xGMI copies where the platform offers them (a refusal only means the copies are staged)
    for (uint32_t j = 0; j < ndev; j++)
      if (M->dev[j] != M->dev[k]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, M->dev[k], M->dev[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(M->dev[j], 0);
      }
    (void)hipGetLastError();
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
  // 1. stage + begin
  int rc = for_each_shard(G, [&](size_t k) {
    int e = g16_stage_witness(m->h[k], 0, wtns, wtns_len);
    if (e) return e;
    uint32_t mask = 0;
    for (uint32_t v = 0; v < 3; v++)
      if (v % G == k) mask |= 1u << v;
    void* outs[3] = {m->vec[0], m->vec[1], m->vec[2]};
    return g16_shard_begin(m->h[k], 0, mask, outs);
  });
  // 2. slices to their shards (the vectors are complete: g16_shard_begin returns after its copies)
  for (size_t k = 0; k < G && !rc; k++) {
    uint32_t lo, hi;
    g16_shard_range(m->N, (int32_t)k, (int32_t)G, &lo, &hi);
    const size_t bytes = (size_t)(hi - lo) * G16_LAZY_FR_BYTES;
    if (bytes && hipSetDevice(m->dev[k]) != hipSuccess) { set_error("hipSetDevice failed"); rc = G16_E_HIP; break; }
    for (uint32_t v = 0; v < 3 && bytes; v++) {
      const int owner = m->dev[v % G];
      const uint8_t* src = (const uint8_t*)m->vec[v] + (size_t)lo * G16_LAZY_FR_BYTES;
      hipError_t e = owner == m->dev[k] ? hipMemcpy(m->slice[k][v], src, bytes, hipMemcpyDeviceToDevice)
                                        : hipMemcpyPeer(m->slice[k][v], m->dev[k], src, owner, bytes);
      if (e != hipSuccess) { set_error(std::string("slice copy failed: ") + hipGetErrorString(e)); rc = G16_E_HIP; break; }
    }
    // A device-to-device hipMemcpy does not wait on the host, and the shards' streams are non-blocking: without this
    // the slice could still be in flight on the null stream when g16_shard_end reads it (r02: seen once the streams'
    // hardware queues were reassigned).  Only the null stream is drained: the witness MSMs keep running.
    if (bytes && !rc && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("slice copy failed"); rc = G16_E_HIP; }
  }
  // 3. join + H-MSM + collect (also run after a failure above, so that every begun shard is drained)
  std::vector<uint8_t> parts(G * G16_PARTIAL_BYTES);
  const bool failed = rc != 0;
  const std::string first_err = failed ? g16_last_error() : "";
  int rc2 = for_each_shard(G, [&](size_t k) {
    const void* sl[3] = {m->slice[k][0], m->slice[k][1], m->slice[k][2]};
    return g16_shard_end(m->h[k], 0, sl, parts.data() + k * G16_PARTIAL_BYTES);
  });
  if (failed) { set_error(first_err); return rc; }
  if (rc2) return rc2;
  // 4. add the partial sums, finish on the host
  return g16_prove_finish(m->h[0], 0, parts.data(), (uint32_t)G, r, s, out, pub);
}

int g16_multi_get_info(const g16_multi* m, g16_info* out, uint32_t* n_shards) {
  if (!m || !out) { set_error("NULL argument"); return G16_E_ARG; }
  if (n_shards) *n_shards = (uint32_t)m->h.size();
  return g16_get_info(m->h[0], out);
}

}  // extern "C"
