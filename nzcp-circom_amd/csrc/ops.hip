// Operator-level entry points: the device twins of ffjavascript's public curve API
// (Fr.fft / Fr.ifft / G.multiExpAffine / frm_mul; pins /root/reference/yarn.lock:408-416,
// 1132-1138).  They exist so the parity tests can pin every layer of the prove path separately
// (field product -> point addition -> NTT -> MSM) through the C ABI.
#include <string.h>

#include <memory>

#include "ec29.cuh"
#include "fr29.cuh"
#include "internal.h"

namespace g16 {

template <class PM>
__global__ void field_op_kernel(const Fp<PM>* a, const Fp<PM>* b, Fp<PM>* out, size_t n, int op) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp<PM> x = a[i], y = b[i], r;
  switch (op) {
    case 0: r = fp_mul(x, y); break;
    case 1: r = fp_add(x, y); break;
    case 2: r = fp_sub(x, y); break;
    case 3: r = fp_to_mont(x); break;
    case 4: r = fp_from_mont(x); break;
    default: r = fp_neg(x); break;
  }
  out[i] = r;
}

// out[i] = a[i] + b[i] as XYZZ (exercises from_affine, madd incl. doubling / inverse / infinity)
template <class F>
__global__ void ec_add_kernel(const Affine<F>* a, const Affine<F>* b, XYZZ<F>* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> acc;
  xyzz_from_affine(acc, a[i]);
  Affine<F> q = b[i];
  if (!aff_is_inf(q)) xyzz_madd(acc, q);
  XYZZ<F> twice = acc;   // also run the full add and the doubling: out = (a+b) + (a+b) - (a+b)... keep simple:
  xyzz_dbl(twice);       // 2(a+b)
  XYZZ<F> neg = acc;
  xyzz_neg(neg);
  xyzz_add(twice, neg);  // 2(a+b) - (a+b) = a+b through add-2008-s
  out[i] = twice;
}

static int dev_check(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device available: libg16hip has no CPU fallback");
    return G16_E_NOGPU;
  }
  if (device < 0 || device >= n) { set_error("device ordinal out of range"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(device));
  return G16_OK;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { G16_HIP(hipMalloc(&p, bytes ? bytes : 16)); return G16_OK; }
};

template <class F> static void to_std_bytes(uint8_t* out, const Affine<F>& p);
template <> void to_std_bytes<FqOps>(uint8_t* out, const G1Affine& p) {
  Fq x = fp_from_mont(p.x), y = fp_from_mont(p.y);
  memcpy(out, x.v, 32); memcpy(out + 32, y.v, 32);
}
template <> void to_std_bytes<Fq2Ops>(uint8_t* out, const G2Affine& p) {
  Fq v[4] = {fp_from_mont(p.x.a), fp_from_mont(p.x.b), fp_from_mont(p.y.a), fp_from_mont(p.y.b)};
  for (int i = 0; i < 4; i++) memcpy(out + 32 * i, v[i].v, 32);
}

template <class F>
static int ec_add_impl(int device, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  int rc = dev_check(device);
  if (rc) return rc;
  DevBuf da, db, dout;
  const size_t ab = n * sizeof(Affine<F>), ob = n * sizeof(XYZZ<F>);
  if ((rc = da.alloc(ab)) || (rc = db.alloc(ab)) || (rc = dout.alloc(ob))) return rc;
  G16_HIP(hipMemcpy(da.p, a, ab, hipMemcpyHostToDevice));
  G16_HIP(hipMemcpy(db.p, b, ab, hipMemcpyHostToDevice));
  ec_add_kernel<F><<<(unsigned)((n + 63) / 64), 64>>>((const Affine<F>*)da.p, (const Affine<F>*)db.p,
                                                      (XYZZ<F>*)dout.p, n);
  G16_HIP(hipGetLastError());
  std::vector<XYZZ<F>> h(n);
  G16_HIP(hipMemcpy(h.data(), dout.p, ob, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++) {
    Affine<F> r;
    xyzz_to_affine(r, h[i]);
    to_std_bytes<F>(out + i * sizeof(Affine<F>), r);
  }
  return G16_OK;
}

template <class F>
static int multiexp_impl(int device, int curve, const uint8_t* bases, const uint8_t* scalars, size_t n,
                         int window_bits, uint8_t* out) {
  int rc = dev_check(device);
  if (rc) return rc;
  if (n >= 0x7fffffffu) { set_error("multiexp: too many points"); return G16_E_ARG; }
  MsmGroup m;
  MsmConfig cfg;
  cfg.c = window_bits;
  MsmWorkspace* ws = nullptr;
  DevBuf ds;
  hipStream_t st = nullptr;
  MsmSectionIn sec;
  if (curve == 2) sec.bases2_host = bases; else sec.bases_host = bases;
  sec.n_total = (uint32_t)n;
  rc = msm_group_create(m, &sec, 1, cfg);
  if (!rc) rc = msm_workspace_create(&ws, m);
  if (!rc) rc = ds.alloc(n * 32);
  if (!rc && hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); rc = G16_E_HIP; }
  if (!rc && hipStreamCreate(&st) != hipSuccess) { set_error("hipStreamCreate failed"); rc = G16_E_HIP; }
  MsmResult res;
  if (!rc) rc = msm_launch(m, ws, (const Fr*)ds.p, st, st);
  if (!rc) rc = msm_collect(m, ws, &res);
  if (!rc) {
    XYZZ<F> total;
    memcpy(&total, curve == 2 ? (const void*)&res.g2 : (const void*)&res.g1[0], sizeof(total));
    Affine<F> r;
    xyzz_to_affine(r, total);
    to_std_bytes<F>(out, r);
  }
  if (st) (void)hipStreamDestroy(st);
  msm_workspace_destroy(ws);
  msm_group_destroy(m);
  return rc;
}

static int fft_impl(int device, uint8_t* buf, size_t n, bool inverse) {
  int rc = dev_check(device);
  if (rc) return rc;
  int L = 0;
  while (((size_t)1 << L) < n) L++;
  if (n == 0 || ((size_t)1 << L) != n) { set_error("fft: size must be a power of two"); return G16_E_ARG; }
  NttTables t;
  DevBuf dx, dy;
  hipStream_t st = nullptr;
  G16_HIP(hipStreamCreate(&st));
  rc = ntt_tables_create(t, L, st);
  if (!rc) rc = dx.alloc(n * 32);
  if (!rc) rc = dy.alloc(n * 32);
  if (!rc && hipMemcpyAsync(dx.p, buf, n * 32, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("hipMemcpy failed"); rc = G16_E_HIP; }
  // dx: canonical image (in, then out); dz: the lazy working vector
  DevBuf dz;
  if (!rc) rc = dz.alloc(n * sizeof(F29));
  F29* vz[1] = {(F29*)dz.p};
  if (!rc) {
    if (inverse) {
      rc = ntt_import(t, (const Fr*)dx.p, (F29*)dz.p, false, st);
      if (!rc) rc = ntt_dif_inverse(t, vz, 1, st);
      if (!rc) rc = ntt_export(t, (const F29*)dz.p, (Fr*)dy.p, true, true, st);
    } else {
      rc = ntt_import(t, (const Fr*)dx.p, (F29*)dz.p, true, st);
      if (!rc) rc = ntt_dit_forward(t, vz, 1, st);
      if (!rc) rc = ntt_export(t, (const F29*)dz.p, (Fr*)dy.p, false, false, st);
    }
  }
  if (!rc && hipMemcpyAsync(buf, dy.p, n * 32, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("hipMemcpy failed"); rc = G16_E_HIP; }
  if (hipStreamSynchronize(st) != hipSuccess && !rc) { set_error("stream sync failed"); rc = G16_E_HIP; }
  ntt_tables_destroy(t);
  (void)hipStreamDestroy(st);
  return rc;
}


// ---------------------------------------------------------------------------------------------------
// Layer tests of the arithmetic the hot path actually runs: the 9 x 29-bit lazy field (fq29.cuh / fr29.cuh)
// and the XYZZ formulas over it (ec29.cuh).  Elements travel as RAW limb images (F29: nine u32 limbs + one
// pad word, value = sum l[i] 2^(29 i)), so a test can place inputs at the documented bounds (X < 5.4p, ...).
template <class C>
__global__ void f29_op_kernel(const F29* a, const F29* b, const F29* c, const F29* d, F29* out, size_t n, int op) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F29 x = a[i], y = b[i], z = c[i], w = d[i];
  F29 r;
  switch (op) {
    case 0: r = f29_mul<C>(x, y); break;
    case 1: r = f29_sqr<C>(x); break;
    case 2: r = f29_mul2<C>(x, y, z, w); break;       // x*y + z*w, one reduction (== Fq29Ops::mul_add)
    case 3: r = f29_sqr_mul<C>(x, z, w); break;       // x^2 + z*w, one reduction
    case 4: r = f29_add<C>(x, y); break;
    case 5: r = f29_sub<2, C>(x, y); break;
    case 6: r = f29_sub<6, C>(x, y); break;
    case 7: r = f29_sub_b_2c<C>(x, y, z); break;      // x + 4p - y - 2z
    case 8: r = f29_neg<2, C>(x); break;
    case 9: r = f29_mul_rows<C>(x, y); break;
    case 10: r = fr29_weak_reduce(x); break;          // Fr only
    case 11: { r = f29_zero(); r.l[0] = f29_is_zero<C>(x) ? 1u : 0u; r.l[1] = f29_maybe_zero<7, C>(x) ? 1u : 0u; break; }
    default: r = f29_zero(); break;
  }
  r.pad_ = 0;
  out[i] = r;
}

// op 0: x29_madd_fast (flag = its redo request), 1: x29_madd, 2: x29_add (q is an XYZZ image), 3: x29_dbl,
// 4: pack -> unpack round trip of the affine q (the resident base format of the accumulate kernel), then madd
template <class F>
__global__ void x29_op_kernel(const XYZZ<F>* acc_in, const void* q_in, XYZZ<F>* out, uint8_t* flags, size_t n, int op) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> acc = acc_in[i];
  bool fl = false;
  if (op == 2) {
    const XYZZ<F> q = reinterpret_cast<const XYZZ<F>*>(q_in)[i];
    x29_add(acc, q);
  } else if (op == 3) {
    x29_dbl(acc);
  } else {
    Affine<F> q = reinterpret_cast<const Affine<F>*>(q_in)[i];
    if (op == 4) {
      PackedAffine<F> pk;
      a29_pack(pk, q);
      a29_unpack(q, pk);
    }
    if (op == 0) fl = x29_madd_fast(acc, q);
    else x29_madd(acc, q);
  }
  out[i] = acc;
  if (flags) flags[i] = fl ? 1 : 0;
}
}  // namespace g16

using namespace g16;

extern "C" {

int g16_fr_fft(int device, uint8_t* buf, size_t n) { return fft_impl(device, buf, n, false); }
int g16_fr_ifft(int device, uint8_t* buf, size_t n) { return fft_impl(device, buf, n, true); }

int g16_field_op(int device, int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  int rc = dev_check(device);
  if (rc) return rc;
  DevBuf da, db, dout;
  if ((rc = da.alloc(n * 32)) || (rc = db.alloc(n * 32)) || (rc = dout.alloc(n * 32))) return rc;
  G16_HIP(hipMemcpy(da.p, a, n * 32, hipMemcpyHostToDevice));
  G16_HIP(hipMemcpy(db.p, b, n * 32, hipMemcpyHostToDevice));
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (field == 0) field_op_kernel<FrParams><<<nb, 256>>>((const Fr*)da.p, (const Fr*)db.p, (Fr*)dout.p, n, op);
  else field_op_kernel<FqParams><<<nb, 256>>>((const Fq*)da.p, (const Fq*)db.p, (Fq*)dout.p, n, op);
  G16_HIP(hipGetLastError());
  G16_HIP(hipMemcpy(out, dout.p, n * 32, hipMemcpyDeviceToHost));
  return G16_OK;
}

int g16_fr_batch_mul(int device, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int field) {
  return g16_field_op(device, field, 0, a, b, out, n);
}

int g16_ec_add(int device, int curve, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  return curve == 2 ? ec_add_impl<Fq2Ops>(device, a, b, out, n) : ec_add_impl<FqOps>(device, a, b, out, n);
}

int g16_g1_multiexp(int device, const uint8_t* bases, const uint8_t* scalars, size_t n, int window_bits,
                    uint8_t out[64]) {
  return multiexp_impl<FqOps>(device, 1, bases, scalars, n, window_bits, out);
}
int g16_g2_multiexp(int device, const uint8_t* bases, const uint8_t* scalars, size_t n, int window_bits,
                    uint8_t out[128]) {
  return multiexp_impl<Fq2Ops>(device, 2, bases, scalars, n, window_bits, out);
}

int g16_f29_op(int device, int field, int op, const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d,
               uint8_t* out, size_t n) {
  int rc = dev_check(device);
  if (rc) return rc;
  if (!a || !out || op < 0 || op > 11 || (op == 10 && field != 0)) { set_error("g16_f29_op: bad argument"); return G16_E_ARG; }
  DevBuf dv[4], dout;
  const uint8_t* src[4] = {a, b ? b : a, c ? c : a, d ? d : a};
  const size_t bytes = n * sizeof(F29);
  for (int k = 0; k < 4; k++) {
    if ((rc = dv[k].alloc(bytes))) return rc;
    G16_HIP(hipMemcpy(dv[k].p, src[k], bytes, hipMemcpyHostToDevice));
  }
  if ((rc = dout.alloc(bytes))) return rc;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (field == 0)
    f29_op_kernel<Fr29C><<<nb, 256>>>((const F29*)dv[0].p, (const F29*)dv[1].p, (const F29*)dv[2].p, (const F29*)dv[3].p, (F29*)dout.p, n, op);
  else
    f29_op_kernel<Fq29C><<<nb, 256>>>((const F29*)dv[0].p, (const F29*)dv[1].p, (const F29*)dv[2].p, (const F29*)dv[3].p, (F29*)dout.p, n, op);
  G16_HIP(hipGetLastError());
  G16_HIP(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
  return G16_OK;
}

int g16_x29_op(int device, int curve, int op, const uint8_t* acc, const uint8_t* q, uint8_t* out, uint8_t* flags, size_t n) {
  int rc = dev_check(device);
  if (rc) return rc;
  if (!acc || !out || op < 0 || op > 4 || (op != 3 && !q)) { set_error("g16_x29_op: bad argument"); return G16_E_ARG; }
  const size_t coord = curve == 2 ? sizeof(F29x2) : sizeof(F29);
  const size_t ab = n * 4 * coord, qb = n * (op == 2 ? 4 : 2) * coord;
  DevBuf da, dq, dout, dfl;
  if ((rc = da.alloc(ab)) || (rc = dq.alloc(qb)) || (rc = dout.alloc(ab)) || (rc = dfl.alloc(n))) return rc;
  G16_HIP(hipMemcpy(da.p, acc, ab, hipMemcpyHostToDevice));
  if (q) G16_HIP(hipMemcpy(dq.p, q, qb, hipMemcpyHostToDevice));
  const unsigned nb = (unsigned)((n + 63) / 64);
  if (curve == 2)
    x29_op_kernel<Fq2x29Ops><<<nb, 64>>>((const G2XYZZ29*)da.p, dq.p, (G2XYZZ29*)dout.p, (uint8_t*)dfl.p, n, op);
  else
    x29_op_kernel<Fq29Ops><<<nb, 64>>>((const G1XYZZ29*)da.p, dq.p, (G1XYZZ29*)dout.p, (uint8_t*)dfl.p, n, op);
  G16_HIP(hipGetLastError());
  G16_HIP(hipMemcpy(out, dout.p, ab, hipMemcpyDeviceToHost));
  if (flags) G16_HIP(hipMemcpy(flags, dfl.p, n, hipMemcpyDeviceToHost));
  return G16_OK;
}

}  // extern "C"
