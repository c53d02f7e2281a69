/*
 * Thin raw-C N-API binding over the C ABI of libg16hip.so (include/g16_prover.h).
 * Host code stays Node.js (BASELINE.json north_star); this file only marshals Buffers.
 *
 * snarkjs never blocks the event loop (its work runs in web-workers: ffjavascript threadman.js,
 * pin /root/reference/yarn.lock:408-416), so create/prove run in napi async work and return
 * Promises; nothing keeps the loop alive after the Promise settles (Threading).
 *
 * Exports:  create(zkey: Buffer, opts: {device, devices: [..], windowBits, taskLen}) -> Promise<handle>
 *             (devices with more than one entry: ONE proof sharded over those GPUs, g16_multi_* -- BASELINE config 4)
 *           prove(handle, wtns: Buffer, r: Buffer|null, s: Buffer|null) -> Promise<{proof: Buffer(256), pub: Buffer}>
 *           proveBatch(handle, wtns: Buffer[], rs: Buffer|null) -> Promise<[{proof, pub}]>
 *           info(handle) -> {nVars, nPublic, domainSize, nCoefs}
 *           timings(handle) -> {...ms}
 *           destroy(handle)
 *           createVerifier(vkey: Buffer, nPublic, montgomery: 0|1, device) -> Promise<vhandle>   (g16_verifier_create)
 *           verifyBatch(vhandle, proofs: Buffer(count*256), pubs: Buffer(count*nPublic*32)) -> Promise<Buffer(count)>
 *           destroyVerifier(vhandle)
 *           plonkCreate(zkey: Buffer, device) -> Promise<phandle>                                  (g16_plonk_create)
 *           plonkProve(phandle, wtns: Buffer, blinding: Buffer(288)|null) -> Promise<{proof: Buffer(832), pub: Buffer}>
 *           plonkDestroy(phandle)
 *           plonkSetupFiles(r1csPath, ptauPath, zkeyPath, device, withLagrange) -> Promise<undefined>   (g16_plonk_setup_files)
 */
#include <node_api.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/g16_prover.h"

#define NAPI_OK(call)                                                         \
  do {                                                                        \
    if ((call) != napi_ok) {                                                  \
      napi_throw_error(env, NULL, "g16 addon: N-API call failed: " #call);    \
      return NULL;                                                            \
    }                                                                         \
  } while (0)

/* `inflight` and `closing` are touched on the main (JS) thread only: jobs are queued by js_prove* and retired by
 * job_complete, both of which run there.  destroy() while a proof is queued or running only marks the handle;
 * the native prover (its mutex, streams, HBM buffers) is released when the last job retires -- never under a
 * worker thread that still dereferences it. */
typedef struct {
  g16_prover* p;   /* one GPU ... */
  g16_multi* m;    /* ... or one proof sharded over several (exactly one of the two is set) */
  uint32_t inflight;
  int closing;
} handle_t;

static int handle_live(const handle_t* h) { return h && (h->p || h->m); }
static int handle_info(const handle_t* h, g16_info* inf) {
  return h->m ? g16_multi_get_info(h->m, inf, NULL) : g16_get_info(h->p, inf);
}

static void handle_release(handle_t* h) {
  if (h->p) g16_destroy(h->p);
  if (h->m) g16_multi_destroy(h->m);
  h->p = NULL;
  h->m = NULL;
}

static void handle_finalize(napi_env env, void* data, void* hint) {
  handle_t* h = (handle_t*)data;   /* no job can be in flight: every job holds a reference on the external */
  handle_release(h);
  free(h);
}

typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref refs[4];      /* keep input Buffers alive while the worker runs */
  int nrefs;
  /* create */
  const uint8_t* zkey; size_t zkey_len; g16_opts opts; g16_prover* created;
  int32_t devices[64]; uint32_t ndev; g16_multi* created_multi;
  /* prove */
  handle_t* h; const uint8_t* wtns; size_t wtns_len; int have_r, have_s; uint8_t r[32], s[32];
  g16_proof proof; uint8_t* pub; size_t pub_len;
  /* batch */
  size_t bcount; const uint8_t** bw; size_t* blen; uint8_t* brs; g16_proof* bproofs; uint8_t* bpub; size_t bpub_each;
  napi_ref* brefs;
  int rc; char err[512];
  int is_create;   /* 1 = create, 0 = prove, 2 = prove batch */
} job_t;

static void job_execute(napi_env env, void* data) {
  job_t* j = (job_t*)data;
  if (j->is_create == 1) {
    if (j->ndev > 1) j->rc = g16_multi_create(j->zkey, j->zkey_len, j->devices, j->ndev, &j->opts, &j->created_multi);
    else j->rc = g16_create(j->zkey, j->zkey_len, &j->opts, &j->created);
  } else if (j->is_create == 2) {
    j->rc = g16_prove_batch(j->h->p, j->bw, j->blen, j->bcount, j->brs, j->bproofs, j->bpub);
  } else {
    if (j->h->m)
      j->rc = g16_multi_prove(j->h->m, j->wtns, j->wtns_len, j->have_r ? j->r : NULL, j->have_s ? j->s : NULL,
                              &j->proof, j->pub);
    else
      j->rc = g16_prove(j->h->p, j->wtns, j->wtns_len, j->have_r ? j->r : NULL, j->have_s ? j->s : NULL,
                        &j->proof, j->pub);
  }
  if (j->rc) {
    strncpy(j->err, g16_last_error(), sizeof(j->err) - 1);   /* thread-local: read on the worker */
    j->err[sizeof(j->err) - 1] = 0;
  }
}

static void job_complete(napi_env env, napi_status status, void* data) {
  job_t* j = (job_t*)data;
  napi_value result;
  if (status != napi_ok || j->rc) {
    napi_value msg, err;
    napi_create_string_utf8(env, j->rc ? j->err : "g16 addon: async work cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
  } else if (j->is_create == 2) {
    napi_create_array_with_length(env, j->bcount, &result);
    for (size_t i = 0; i < j->bcount; i++) {
      napi_value item, proof, pub;
      void* dst;
      napi_create_object(env, &item);
      napi_create_buffer_copy(env, sizeof(g16_proof), &j->bproofs[i], &dst, &proof);
      napi_create_buffer_copy(env, j->bpub_each, j->bpub + i * j->bpub_each, &dst, &pub);
      napi_set_named_property(env, item, "proof", proof);
      napi_set_named_property(env, item, "pub", pub);
      napi_set_element(env, result, (uint32_t)i, item);
    }
    napi_resolve_deferred(env, j->deferred, result);
  } else if (j->is_create == 1) {
    handle_t* h = (handle_t*)calloc(1, sizeof(handle_t));
    h->p = j->created;
    h->m = j->created_multi;
    napi_create_external(env, h, handle_finalize, NULL, &result);
    napi_resolve_deferred(env, j->deferred, result);
  } else {
    napi_value proof, pub;
    void* dst;
    napi_create_object(env, &result);
    napi_create_buffer_copy(env, sizeof(g16_proof), &j->proof, &dst, &proof);
    napi_create_buffer_copy(env, j->pub_len, j->pub, &dst, &pub);
    napi_set_named_property(env, result, "proof", proof);
    napi_set_named_property(env, result, "pub", pub);
    napi_resolve_deferred(env, j->deferred, result);
  }
  if (j->h) {   /* prove / proveBatch: retire the job; a destroy() issued meanwhile takes effect now */
    j->h->inflight--;
    if (j->h->closing && j->h->inflight == 0) handle_release(j->h);
  }
  for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
  if (j->brefs) { for (size_t i = 0; i < j->bcount; i++) napi_delete_reference(env, j->brefs[i]); free(j->brefs); }
  napi_delete_async_work(env, j->work);
  free(j->pub); free(j->bw); free(j->blen); free(j->brs); free(j->bproofs); free(j->bpub);
  free(j);
}

static int32_t get_i32(napi_env env, napi_value obj, const char* key, int32_t dflt) {
  napi_value v;
  napi_valuetype t;
  bool has = false;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return dflt;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return dflt;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return dflt;
  int32_t out = dflt;
  napi_get_value_int32(env, v, &out);
  return out;
}

static napi_value queue_job(napi_env env, job_t* j, const char* name) {
  napi_value promise, resname;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  NAPI_OK(napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &resname));
  NAPI_OK(napi_create_async_work(env, NULL, resname, job_execute, job_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}

static napi_value js_create(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  bool isbuf = false;
  if (argc < 1 || napi_is_buffer(env, argv[0], &isbuf) != napi_ok || !isbuf) {
    napi_throw_type_error(env, NULL, "create(zkey: Buffer, opts)");
    return NULL;
  }
  job_t* j = (job_t*)calloc(1, sizeof(job_t));
  j->is_create = 1;
  void* data;
  NAPI_OK(napi_get_buffer_info(env, argv[0], &data, &j->zkey_len));
  j->zkey = (const uint8_t*)data;
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->refs[j->nrefs++]));
  if (argc > 1) {
    napi_valuetype t;
    if (napi_typeof(env, argv[1], &t) == napi_ok && t == napi_object) {
      j->opts.device = get_i32(env, argv[1], "device", 0);
      j->opts.window_bits = get_i32(env, argv[1], "windowBits", 0);
      j->opts.task_len = get_i32(env, argv[1], "taskLen", 0);
      /* devices: [ordinal, ...] -> one shard of the key per entry; a single entry is the plain one-GPU handle */
      napi_value dv;
      bool has = false, isarr = false;
      if (napi_has_named_property(env, argv[1], "devices", &has) == napi_ok && has &&
          napi_get_named_property(env, argv[1], "devices", &dv) == napi_ok && napi_is_array(env, dv, &isarr) == napi_ok && isarr) {
        uint32_t n = 0;
        napi_get_array_length(env, dv, &n);
        if (n > 64) { free(j); napi_throw_range_error(env, NULL, "devices: at most 64 entries"); return NULL; }
        for (uint32_t i = 0; i < n; i++) {
          napi_value el;
          int32_t d = 0;
          if (napi_get_element(env, dv, i, &el) != napi_ok || napi_get_value_int32(env, el, &d) != napi_ok) {
            free(j);
            napi_throw_type_error(env, NULL, "devices: array of integers expected");
            return NULL;
          }
          j->devices[i] = d;
        }
        j->ndev = n;
        if (n == 1) j->opts.device = j->devices[0];
      }
    }
  }
  return queue_job(env, j, "g16_create");
}

static napi_value js_prove(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  handle_t* h = NULL;
  bool isbuf = false;
  if (argc < 2 || napi_get_value_external(env, argv[0], (void**)&h) != napi_ok || !handle_live(h) || h->closing ||
      napi_is_buffer(env, argv[1], &isbuf) != napi_ok || !isbuf) {
    napi_throw_type_error(env, NULL, "prove(handle, wtns: Buffer, r, s): bad arguments or handle already destroyed");
    return NULL;
  }
  job_t* j = (job_t*)calloc(1, sizeof(job_t));
  j->h = h;
  void* data;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &j->wtns_len));
  j->wtns = (const uint8_t*)data;
  NAPI_OK(napi_create_reference(env, argv[1], 1, &j->refs[j->nrefs++]));
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->refs[j->nrefs++]));
  for (int k = 0; k < 2; k++) {
    if (argc > (size_t)(2 + k) && napi_is_buffer(env, argv[2 + k], &isbuf) == napi_ok && isbuf) {
      size_t len;
      NAPI_OK(napi_get_buffer_info(env, argv[2 + k], &data, &len));
      if (len != 32) { free(j); napi_throw_range_error(env, NULL, "r and s must be 32-byte Buffers"); return NULL; }
      memcpy(k == 0 ? j->r : j->s, data, 32);
      if (k == 0) j->have_r = 1; else j->have_s = 1;
    }
  }
  g16_info inf;
  handle_info(h, &inf);
  j->pub_len = (size_t)inf.n_public * 32;
  j->pub = (uint8_t*)malloc(j->pub_len ? j->pub_len : 1);
  h->inflight++;
  return queue_job(env, j, "g16_prove");
}

/* proveBatch(handle, wtns: Buffer[], rs: Buffer(count*64)|null) -> Promise<[{proof, pub}]>  (g16_prove_batch:
 * the library pipelines the proofs over two scratch contexts) */
static napi_value js_prove_batch(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  handle_t* h = NULL;
  bool isarr = false, isbuf = false;
  uint32_t n = 0;
  if (argc < 2 || napi_get_value_external(env, argv[0], (void**)&h) != napi_ok || !h || !h->p || h->closing ||
      napi_is_array(env, argv[1], &isarr) != napi_ok || !isarr || napi_get_array_length(env, argv[1], &n) != napi_ok || n == 0) {
    napi_throw_type_error(env, NULL, "proveBatch(handle, wtns: Buffer[], rs): bad arguments, destroyed handle, or a "
                                     "multi-device handle (batches run on one-GPU handles: replicas, SURVEY 8e)");
    return NULL;
  }
  job_t* j = (job_t*)calloc(1, sizeof(job_t));
  j->is_create = 2;
  j->bcount = n;
  j->bw = (const uint8_t**)calloc(n, sizeof(uint8_t*));
  j->blen = (size_t*)calloc(n, sizeof(size_t));
  j->brefs = (napi_ref*)calloc(n, sizeof(napi_ref));
  j->bproofs = (g16_proof*)calloc(n, sizeof(g16_proof));
  g16_info inf;
  g16_get_info(h->p, &inf);
  j->bpub_each = (size_t)inf.n_public * 32;
  j->bpub = (uint8_t*)malloc(j->bpub_each * n + 1);
  for (uint32_t i = 0; i < n; i++) {
    napi_value el;
    void* data;
    if (napi_get_element(env, argv[1], i, &el) != napi_ok || napi_is_buffer(env, el, &isbuf) != napi_ok || !isbuf ||
        napi_get_buffer_info(env, el, &data, &j->blen[i]) != napi_ok || napi_create_reference(env, el, 1, &j->brefs[i]) != napi_ok) {
      napi_throw_type_error(env, NULL, "proveBatch: every witness must be a Buffer");
      return NULL;   /* small leak on a caller bug; the job never ran */
    }
    j->bw[i] = (const uint8_t*)data;
  }
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->refs[j->nrefs++]));
  if (argc > 2 && napi_is_buffer(env, argv[2], &isbuf) == napi_ok && isbuf) {
    void* data;
    size_t len;
    NAPI_OK(napi_get_buffer_info(env, argv[2], &data, &len));
    if (len != (size_t)n * 64) { napi_throw_range_error(env, NULL, "rs must hold count*64 bytes"); return NULL; }
    j->brs = (uint8_t*)malloc(len);
    memcpy(j->brs, data, len);
  }
  j->h = h;   /* set last: the early error returns above must not retire a job that was never counted */
  h->inflight++;
  return queue_job(env, j, "g16_prove_batch");
}

static napi_value js_info(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out, v;
  handle_t* h = NULL;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || napi_get_value_external(env, argv[0], (void**)&h) != napi_ok || !handle_live(h)) {
    napi_throw_type_error(env, NULL, "info(handle)");
    return NULL;
  }
  g16_info inf;
  handle_info(h, &inf);
  NAPI_OK(napi_create_object(env, &out));
  const char* keys[4] = {"nVars", "nPublic", "domainSize", "nCoefs"};
  uint32_t vals[4] = {inf.n_vars, inf.n_public, inf.domain_size, inf.n_coefs};
  for (int i = 0; i < 4; i++) {
    NAPI_OK(napi_create_uint32(env, vals[i], &v));
    NAPI_OK(napi_set_named_property(env, out, keys[i], v));
  }
  return out;
}

static napi_value js_timings(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out, v;
  handle_t* h = NULL;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || napi_get_value_external(env, argv[0], (void**)&h) != napi_ok || !h || !h->p) {
    napi_throw_type_error(env, NULL, "timings(handle)");
    return NULL;
  }
  g16_timings t;
  g16_get_timings(h->p, &t);
  NAPI_OK(napi_create_object(env, &out));
  const char* keys[4] = {"uploadMs", "qapMs", "nttMs", "totalMs"};
  double vals[4] = {t.upload_ms, t.qap_ms, t.ntt_ms, t.total_ms};
  for (int i = 0; i < 4; i++) {
    NAPI_OK(napi_create_double(env, vals[i], &v));
    NAPI_OK(napi_set_named_property(env, out, keys[i], v));
  }
  return out;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  handle_t* h = NULL;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc >= 1 && napi_get_value_external(env, argv[0], (void**)&h) == napi_ok && handle_live(h)) {
    h->closing = 1;                       /* no new jobs; info()/timings() stay valid until the release */
    if (h->inflight == 0) handle_release(h);
  }
  return NULL;
}

/* ------------------------------------------------------------------ verifier (g16_verifier_*): snarkjs groth16.verify */
typedef struct {
  g16_verifier* v;            /* Groth16 ... */
  g16_plonk_verifier* pv;     /* ... or PLONK (exactly one of the two is set) */
  uint32_t n_public;
  uint32_t inflight;
  int closing;
} vhandle_t;
static void vhandle_release(vhandle_t* h) {
  if (h->v) g16_verifier_destroy(h->v);
  if (h->pv) g16_plonk_verifier_destroy(h->pv);
  h->v = NULL;
  h->pv = NULL;
}

static void vhandle_finalize(napi_env env, void* data, void* hint) {
  vhandle_t* h = (vhandle_t*)data;
  vhandle_release(h);
  free(h);
}

typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref refs[3];
  int nrefs;
  int kind;                 /* 0 = create, 1 = verify */
  int plonk;                /* the PLONK verifier (vkey = the 712-byte image, proofs of 832 bytes) */
  const uint8_t* vkey; size_t vkey_len; uint32_t n_public; int montgomery, device; g16_verifier* created;
  g16_plonk_verifier* created_pv;
  vhandle_t* h; const uint8_t* proofs; const uint8_t* pubs; size_t count; uint8_t* ok;
  int rc; char err[512];
} vjob_t;

static void vjob_execute(napi_env env, void* data) {
  vjob_t* j = (vjob_t*)data;
  if (j->plonk) {
    if (j->kind == 0) j->rc = g16_plonk_verifier_create(j->vkey, j->vkey_len, j->device, &j->created_pv);
    else j->rc = g16_plonk_verify_batch(j->h->pv, (const g16_plonk_proof*)j->proofs, j->pubs, j->count, j->ok);
  } else if (j->kind == 0) {
    j->rc = g16_verifier_create(j->vkey, j->vkey_len, j->n_public, j->montgomery, j->device, &j->created);
  } else {
    j->rc = g16_verify_batch(j->h->v, (const g16_proof*)j->proofs, j->pubs, j->count, j->ok);
  }
  if (j->rc) {
    strncpy(j->err, g16_last_error(), sizeof(j->err) - 1);
    j->err[sizeof(j->err) - 1] = 0;
  }
}

static void vjob_complete(napi_env env, napi_status status, void* data) {
  vjob_t* j = (vjob_t*)data;
  napi_value result;
  if (status != napi_ok || j->rc) {
    napi_value msg, err;
    napi_create_string_utf8(env, j->rc ? j->err : "g16 addon: async work cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
  } else if (j->kind == 0) {
    vhandle_t* h = (vhandle_t*)calloc(1, sizeof(vhandle_t));
    h->v = j->created;
    h->pv = j->created_pv;
    h->n_public = j->n_public;
    napi_create_external(env, h, vhandle_finalize, NULL, &result);
    napi_resolve_deferred(env, j->deferred, result);
  } else {
    void* dst;
    napi_create_buffer_copy(env, j->count, j->ok, &dst, &result);
    napi_resolve_deferred(env, j->deferred, result);
  }
  if (j->h) {
    j->h->inflight--;
    if (j->h->closing && j->h->inflight == 0) vhandle_release(j->h);
  }
  for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
  napi_delete_async_work(env, j->work);
  free(j->ok);
  free(j);
}

static napi_value queue_vjob(napi_env env, vjob_t* j, const char* name) {
  napi_value promise, resname;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  NAPI_OK(napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &resname));
  NAPI_OK(napi_create_async_work(env, NULL, resname, vjob_execute, vjob_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}

static napi_value js_create_verifier(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  bool isbuf = false;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 4 || napi_is_buffer(env, argv[0], &isbuf) != napi_ok || !isbuf) {
    napi_throw_type_error(env, NULL, "createVerifier(vkey: Buffer, nPublic, montgomery, device[, plonk])");
    return NULL;
  }
  vjob_t* j = (vjob_t*)calloc(1, sizeof(vjob_t));
  void* data;
  int32_t mont = 0, dev = 0, pl = 0;
  if (argc > 4) napi_get_value_int32(env, argv[4], &pl);
  j->plonk = pl != 0;
  NAPI_OK(napi_get_buffer_info(env, argv[0], &data, &j->vkey_len));
  j->vkey = (const uint8_t*)data;
  NAPI_OK(napi_get_value_uint32(env, argv[1], &j->n_public));
  NAPI_OK(napi_get_value_int32(env, argv[2], &mont));
  NAPI_OK(napi_get_value_int32(env, argv[3], &dev));
  j->montgomery = mont;
  j->device = dev;
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->refs[j->nrefs++]));
  return queue_vjob(env, j, "g16_verifier_create");
}

static napi_value js_verify_batch(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  vhandle_t* h = NULL;
  bool b1 = false, b2 = false;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 3 || napi_get_value_external(env, argv[0], (void**)&h) != napi_ok || !h || (!h->v && !h->pv) || h->closing ||
      napi_is_buffer(env, argv[1], &b1) != napi_ok || !b1 || napi_is_buffer(env, argv[2], &b2) != napi_ok || !b2) {
    napi_throw_type_error(env, NULL, "verifyBatch(vhandle, proofs: Buffer, pubs: Buffer)");
    return NULL;
  }
  void *pd, *ud;
  size_t plen, ulen;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &pd, &plen));
  NAPI_OK(napi_get_buffer_info(env, argv[2], &ud, &ulen));
  const size_t psz = h->pv ? sizeof(g16_plonk_proof) : sizeof(g16_proof);
  if (plen % psz || ulen != (plen / psz) * (size_t)h->n_public * 32) {
    napi_throw_range_error(env, NULL, "verifyBatch: proofs must be count*256 (groth16) / count*832 (plonk) bytes and pubs count*nPublic*32 bytes");
    return NULL;
  }
  vjob_t* j = (vjob_t*)calloc(1, sizeof(vjob_t));
  j->kind = 1;
  j->plonk = h->pv != NULL;
  j->h = h;
  j->proofs = (const uint8_t*)pd;
  j->pubs = (const uint8_t*)ud;
  j->count = plen / psz;
  j->ok = (uint8_t*)calloc(j->count ? j->count : 1, 1);
  h->inflight++;
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->refs[j->nrefs++]));
  NAPI_OK(napi_create_reference(env, argv[1], 1, &j->refs[j->nrefs++]));
  NAPI_OK(napi_create_reference(env, argv[2], 1, &j->refs[j->nrefs++]));
  return queue_vjob(env, j, "g16_verify_batch");
}

static napi_value js_destroy_verifier(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  vhandle_t* h = NULL;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc >= 1 && napi_get_value_external(env, argv[0], (void**)&h) == napi_ok && h && (h->v || h->pv)) {
    h->closing = 1;
    if (h->inflight == 0) vhandle_release(h);
  }
  return NULL;
}

/* ------------------------------------------------------------------ PLONK prover (g16_plonk_*): snarkjs plonk.prove */
typedef struct {
  g16_plonk* p;
  uint32_t n_public;
  uint32_t inflight;
  int closing;
} phandle_t;

static void phandle_finalize(napi_env env, void* data, void* hint) {
  phandle_t* h = (phandle_t*)data;
  if (h->p) g16_plonk_destroy(h->p);
  free(h);
}

typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref refs[3];
  int nrefs;
  int kind;                 /* 0 = create, 1 = prove */
  const uint8_t* zkey; size_t zkey_len; int device; g16_plonk* created; uint32_t info[6];
  phandle_t* h; const uint8_t* wtns; size_t wtns_len; int have_blind; uint8_t blind[288];
  g16_plonk_proof proof; uint8_t* pub; size_t pub_len;
  int rc; char err[512];
} pjob_t;

static void pjob_execute(napi_env env, void* data) {
  pjob_t* j = (pjob_t*)data;
  if (j->kind == 0) {
    j->rc = g16_plonk_create(j->zkey, j->zkey_len, j->device, &j->created);
    if (!j->rc) g16_plonk_get_info(j->created, j->info);
  } else {
    j->rc = g16_plonk_prove(j->h->p, j->wtns, j->wtns_len, j->have_blind ? j->blind : NULL, &j->proof, j->pub);
  }
  if (j->rc) {
    strncpy(j->err, g16_last_error(), sizeof(j->err) - 1);
    j->err[sizeof(j->err) - 1] = 0;
  }
}

static void pjob_complete(napi_env env, napi_status status, void* data) {
  pjob_t* j = (pjob_t*)data;
  napi_value result;
  if (status != napi_ok || j->rc) {
    napi_value msg, err;
    napi_create_string_utf8(env, j->rc ? j->err : "g16 addon: async work cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
  } else if (j->kind == 0) {
    phandle_t* h = (phandle_t*)calloc(1, sizeof(phandle_t));
    h->p = j->created;
    h->n_public = j->info[1];
    napi_create_external(env, h, phandle_finalize, NULL, &result);
    napi_resolve_deferred(env, j->deferred, result);
  } else {
    napi_value proof, pub;
    void* dst;
    napi_create_object(env, &result);
    napi_create_buffer_copy(env, sizeof(g16_plonk_proof), &j->proof, &dst, &proof);
    napi_create_buffer_copy(env, j->pub_len, j->pub, &dst, &pub);
    napi_set_named_property(env, result, "proof", proof);
    napi_set_named_property(env, result, "pub", pub);
    napi_resolve_deferred(env, j->deferred, result);
  }
  if (j->h) {
    j->h->inflight--;
    if (j->h->closing && j->h->inflight == 0 && j->h->p) { g16_plonk_destroy(j->h->p); j->h->p = NULL; }
  }
  for (int i = 0; i < j->nrefs; i++) napi_delete_reference(env, j->refs[i]);
  napi_delete_async_work(env, j->work);
  free(j->pub);
  free(j);
}

static napi_value queue_pjob(napi_env env, pjob_t* j, const char* name) {
  napi_value promise, resname;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  NAPI_OK(napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &resname));
  NAPI_OK(napi_create_async_work(env, NULL, resname, pjob_execute, pjob_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}

static napi_value js_plonk_create(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  bool isbuf = false;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || napi_is_buffer(env, argv[0], &isbuf) != napi_ok || !isbuf) {
    napi_throw_type_error(env, NULL, "plonkCreate(zkey: Buffer, device)");
    return NULL;
  }
  pjob_t* j = (pjob_t*)calloc(1, sizeof(pjob_t));
  void* data;
  int32_t dev = 0;
  NAPI_OK(napi_get_buffer_info(env, argv[0], &data, &j->zkey_len));
  j->zkey = (const uint8_t*)data;
  if (argc > 1) napi_get_value_int32(env, argv[1], &dev);
  j->device = dev;
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->refs[j->nrefs++]));
  return queue_pjob(env, j, "g16_plonk_create");
}

static napi_value js_plonk_prove(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  phandle_t* h = NULL;
  bool isbuf = false;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || napi_get_value_external(env, argv[0], (void**)&h) != napi_ok || !h || !h->p || h->closing ||
      napi_is_buffer(env, argv[1], &isbuf) != napi_ok || !isbuf) {
    napi_throw_type_error(env, NULL, "plonkProve(phandle, wtns: Buffer, blinding: Buffer|null)");
    return NULL;
  }
  pjob_t* j = (pjob_t*)calloc(1, sizeof(pjob_t));
  j->kind = 1;
  j->h = h;
  void* data;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &j->wtns_len));
  j->wtns = (const uint8_t*)data;
  if (argc > 2) {
    bool b = false;
    size_t bl = 0;
    void* bd;
    if (napi_is_buffer(env, argv[2], &b) == napi_ok && b) {
      if (napi_get_buffer_info(env, argv[2], &bd, &bl) != napi_ok || bl != 288) {
        free(j);
        napi_throw_range_error(env, NULL, "plonkProve: blinding must be nine 32-byte scalars (288 bytes)");
        return NULL;
      }
      memcpy(j->blind, bd, 288);
      j->have_blind = 1;
    }
  }
  j->pub_len = (size_t)h->n_public * 32;
  j->pub = (uint8_t*)calloc(j->pub_len ? j->pub_len : 1, 1);
  h->inflight++;
  NAPI_OK(napi_create_reference(env, argv[0], 1, &j->refs[j->nrefs++]));
  NAPI_OK(napi_create_reference(env, argv[1], 1, &j->refs[j->nrefs++]));
  return queue_pjob(env, j, "g16_plonk_prove");
}

static napi_value js_plonk_destroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  phandle_t* h = NULL;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc >= 1 && napi_get_value_external(env, argv[0], (void**)&h) == napi_ok && h && h->p) {
    h->closing = 1;
    if (h->inflight == 0) { g16_plonk_destroy(h->p); h->p = NULL; }
  }
  return NULL;
}

/* plonk setup from / to files (the inputs may exceed a Buffer) */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  char r1cs[1024], ptau[1024], zkey[1024];
  int device, lagrange, rc;
  char err[512];
} sjob_t;
static void sjob_execute(napi_env env, void* data) {
  sjob_t* j = (sjob_t*)data;
  j->rc = g16_plonk_setup_files(j->r1cs, j->ptau, j->zkey, j->device, j->lagrange);
  if (j->rc) { strncpy(j->err, g16_last_error(), sizeof(j->err) - 1); j->err[sizeof(j->err) - 1] = 0; }
}
static void sjob_complete(napi_env env, napi_status status, void* data) {
  sjob_t* j = (sjob_t*)data;
  if (status != napi_ok || j->rc) {
    napi_value msg, err;
    napi_create_string_utf8(env, j->rc ? j->err : "g16 addon: async work cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
  } else {
    napi_value undef;
    napi_get_undefined(env, &undef);
    napi_resolve_deferred(env, j->deferred, undef);
  }
  napi_delete_async_work(env, j->work);
  free(j);
}
static napi_value js_plonk_setup_files(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5], promise, resname;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 3) { napi_throw_type_error(env, NULL, "plonkSetupFiles(r1csPath, ptauPath, zkeyPath, device, withLagrange)"); return NULL; }
  sjob_t* j = (sjob_t*)calloc(1, sizeof(sjob_t));
  size_t n = 0;
  if (napi_get_value_string_utf8(env, argv[0], j->r1cs, sizeof(j->r1cs), &n) != napi_ok ||
      napi_get_value_string_utf8(env, argv[1], j->ptau, sizeof(j->ptau), &n) != napi_ok ||
      napi_get_value_string_utf8(env, argv[2], j->zkey, sizeof(j->zkey), &n) != napi_ok) {
    free(j);
    napi_throw_type_error(env, NULL, "plonkSetupFiles: three path strings expected");
    return NULL;
  }
  int32_t v = 0;
  if (argc > 3 && napi_get_value_int32(env, argv[3], &v) == napi_ok) j->device = v;
  v = 0;
  if (argc > 4 && napi_get_value_int32(env, argv[4], &v) == napi_ok) j->lagrange = v;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  NAPI_OK(napi_create_string_utf8(env, "g16_plonk_setup_files", NAPI_AUTO_LENGTH, &resname));
  NAPI_OK(napi_create_async_work(env, NULL, resname, sjob_execute, sjob_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}

static napi_value init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"create", NULL, js_create, NULL, NULL, NULL, napi_default, NULL},
      {"prove", NULL, js_prove, NULL, NULL, NULL, napi_default, NULL},
      {"proveBatch", NULL, js_prove_batch, NULL, NULL, NULL, napi_default, NULL},
      {"info", NULL, js_info, NULL, NULL, NULL, napi_default, NULL},
      {"timings", NULL, js_timings, NULL, NULL, NULL, napi_default, NULL},
      {"destroy", NULL, js_destroy, NULL, NULL, NULL, napi_default, NULL},
      {"createVerifier", NULL, js_create_verifier, NULL, NULL, NULL, napi_default, NULL},
      {"verifyBatch", NULL, js_verify_batch, NULL, NULL, NULL, napi_default, NULL},
      {"destroyVerifier", NULL, js_destroy_verifier, NULL, NULL, NULL, napi_default, NULL},
      {"plonkCreate", NULL, js_plonk_create, NULL, NULL, NULL, napi_default, NULL},
      {"plonkProve", NULL, js_plonk_prove, NULL, NULL, NULL, napi_default, NULL},
      {"plonkDestroy", NULL, js_plonk_destroy, NULL, NULL, NULL, napi_default, NULL},
      {"plonkSetupFiles", NULL, js_plonk_setup_files, NULL, NULL, NULL, napi_default, NULL},
  };
  NAPI_OK(napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props));
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
