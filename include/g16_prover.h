/*
 * g16_prover.h -- C ABI of libg16hip.so, the MI355X (gfx950) Groth16 prover for snarkjs-format
 * proving keys.  This is the drop-in boundary: plain pointers and sizes,
 * no C++ or torch types, no exceptions across the boundary.
 *
 * What each entry point replaces.  The reference repo has no prover source; the path lives in
 * the npm packages it pins (never vendored):
 *     snarkjs 0.4.12        /root/reference/yarn.lock:987-1001   (package.json:12)
 *     ffjavascript 0.2.48   /root/reference/yarn.lock:408-416
 *     wasmcurves 0.1.0      /root/reference/yarn.lock:1132-1138
 *     @iden3/binfileutils   /root/reference/yarn.lock:10-16
 * and the only snarkjs call sites in the reference are the CLI lines /root/reference/Makefile:30-33.
 * The [EXT] names below are the published functions of those packages.
 *
 * Conventions
 *   - return 0 on success, negative G16_E_* on failure; g16_last_error() gives the text
 *     (thread-local; for input errors the text is snarkjs's own Error message).
 *   - the caller owns every buffer it passes; the library copies what it keeps.
 *   - field elements are 32 bytes little-endian.  Proof points come out affine in STANDARD
 *     (non-Montgomery) form: that is what snarkjs `G1.toObject` stringifies.
 *   - one in-flight call per g16_prover handle; different handles may be used concurrently.
 */
#ifndef G16_PROVER_H
#define G16_PROVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G16_OK 0
#define G16_E_ARG (-1)    /* bad argument */
#define G16_E_FORMAT (-2) /* malformed zkey/wtns (message = snarkjs's) */
#define G16_E_HIP (-3)    /* HIP runtime error */
#define G16_E_NOGPU (-4)  /* no gfx950 device / extension cannot run: never falls back to CPU */
#define G16_E_STATE (-5)

typedef struct g16_prover g16_prover;

typedef struct g16_opts {
  int32_t device;       /* HIP device ordinal */
  int32_t shard_rank;   /* this handle owns point range [rank*n/count, (rank+1)*n/count) of every */
  int32_t shard_count;  /* base section; 0/1 = whole key                               */
  int32_t window_bits;  /* Pippenger window c (0 = auto)                                           */
  int32_t task_len;     /* max sorted entries per bucket-accumulation task (0 = auto)              */
  uint32_t flags;       /* bits 8..15: MSM window-precomputation factor (0 = default, see G16_OPT_PRECOMP); other bits 0 */
} g16_opts;
/* flags value asking for a base table with 2^(c*W*k)*P for k < factor (more HBM, fewer bucket rows) */
#define G16_OPT_PRECOMP(factor) (((uint32_t)(factor) & 0xffu) << 8)

typedef struct g16_proof {
  uint8_t a[64];  /* pi_a: x | y                      */
  uint8_t b[128]; /* pi_b: x.c0 | x.c1 | y.c0 | y.c1  */
  uint8_t c[64];  /* pi_c: x | y                      */
} g16_proof;      /* infinity = all-zero bytes        */

typedef struct g16_info {
  uint32_t n_vars, n_public, domain_size, n_coefs;
  uint32_t n_a, n_b1, n_b2, n_c, n_h; /* non-infinity bases resident per MSM (this shard) */
  uint32_t window_bits[5];            /* c chosen for A, B1, B2, C, H */
} g16_info;

/* Per-phase device timings of the last g16_prove* on this handle, milliseconds (HIP events on the
 * prover's streams).  The witness MSMs A, B1, C share one front end and one G1 bucket-accumulate launch, B2 rides
 * on B1's buckets as the G2 lane: msm_ms[0] = that group's main chain, msm_ms[2] = until the G2 lane ends (traced
 * runs only), msm_ms[4] = the H-MSM, [1] and [3] are 0.  *_kernel_ms are the bucket-accumulate kernels alone
 * (the roofline kernel of bench.py): [0] = G1 over A+B1+C, [2] = G2 (B2), [4] = H. */
typedef struct g16_timings {
  float upload_ms, qap_ms, ntt_ms, msm_ms[5], tail_ms, total_ms;
  float msm_accum_kernel_ms[5];
} g16_timings;

/* [EXT] snarkjs groth16_prove.js: readBinFile(zkey,"zkey",2) + zkey_utils.readHeader + the
 * section reads 4..9 of groth16.prove.  Parses and validates, regroups section 4 into CSR,
 * uploads coefficients and (this shard's) bases to HBM once, builds NTT tables.
 * Errors: "<name>: Invalid File format", "Version not supported", "zkey file is not groth16". */
int g16_create(const uint8_t* zkey, size_t zkey_len, const g16_opts* opts, g16_prover** out);

/* [EXT] snarkjs groth16.prove(zkey, wtns): buildABC1 -> 3x(Fr.ifft, batchApplyKey, Fr.fft) ->
 * joinABC -> 5x multiExpAffine -> blinding -> toAffine.  r, s: 32-byte LE scalars < r, or NULL
 * for the OS CSPRNG (snarkjs: Fr.random()).  pub receives n_public * 32 bytes (w[1..p]).
 * Errors: "Curve of the witness does not match the curve of the proving key",
 *         "Invalid witness length. Circuit: N, witness: M". */
int g16_prove(g16_prover* p, const uint8_t* wtns, size_t wtns_len, const uint8_t r[32],
              const uint8_t s[32], g16_proof* out, uint8_t* pub);

/* Batch of independent witnesses against the resident key (BASELINE config 3).  rs: count*64
 * bytes (r|s per proof) or NULL; pub: count*n_public*32 bytes or NULL. */
int g16_prove_batch(g16_prover* p, const uint8_t* const* wtns, const size_t* wtns_lens, size_t count,
                    const uint8_t* rs, g16_proof* out, uint8_t* pub);

/* HBM-resident witness slots: g16_stage_witness parses + uploads (the PCIe leg), g16_prove_staged
 * runs the device pipeline on a staged slot -- bench.py times the latter ("inputs already
 * resident in HBM").  Slots are created on demand; staging the same slot again overwrites it. */
int g16_stage_witness(g16_prover* p, uint32_t slot, const uint8_t* wtns, size_t wtns_len);
int g16_prove_staged(g16_prover* p, uint32_t slot, const uint8_t r[32], const uint8_t s[32],
                     g16_proof* out, uint8_t* pub);

/* Multi-GPU: a sharded handle computes its partial MSM sums; the host exchanges the
 * G16_PARTIAL_BYTES blobs (RCCL all-gather of bytes) and any rank finishes the proof.
 * partials = count blobs back to back, one per shard rank. */
#define G16_PARTIAL_BYTES (128 * 4 + 256) /* XYZZ sums A,B1,C,H (G1) + B2 (G2), Montgomery */
int g16_prove_partial(g16_prover* p, uint32_t slot, uint8_t partial[G16_PARTIAL_BYTES]);
int g16_prove_finish(g16_prover* p, uint32_t slot, const uint8_t* partials, uint32_t count,
                     const uint8_t r[32], const uint8_t s[32], g16_proof* out, uint8_t* pub);

/* Multi-GPU with the H-polynomial pipeline sharded as well (SURVEY 8e "later: split A/B/C across GPUs"): instead
 * of every shard repeating QAP + 6 NTTs, shard v mod count evaluates vector v (0 = A, 1 = B, 2 = C) on the odd
 * coset and the shards exchange slices, so that each joins and multi-exponentiates only its own range of P:
 *   g16_shard_begin  starts this shard's witness MSMs (they keep running) and, for every bit v of vec_mask,
 *                    computes the coset evaluations of vector v and copies all domain_size elements to
 *                    out_vecs[v] -- any memory the device can address (device, peer device, pinned or plain host);
 *                    returns when the copies are complete.  Elements are G16_LAZY_FR_BYTES each (the kernels'
 *                    9 x 29-bit lazy Montgomery image: opaque, only to be handed to g16_shard_end).
 *   (exchange)       shard r needs elements [lo, hi) = g16_shard_range(domain_size, r, count) of all three vectors:
 *                    an RCCL scatter / all-to-all between processes, peer copies inside one process.
 *   g16_shard_end    takes the three slices ((hi - lo) elements each), joins P = A.B - C on the slice, runs the
 *                    H-MSM over its bases and returns the partial sums exactly like g16_prove_partial. */
#define G16_LAZY_FR_BYTES 40
int g16_shard_begin(g16_prover* p, uint32_t slot, uint32_t vec_mask, void* const out_vecs[3]);
int g16_shard_end(g16_prover* p, uint32_t slot, const void* const slices[3], uint8_t partial[G16_PARTIAL_BYTES]);

/* One PROCESS, several GPUs (the Node.js host of BASELINE config 4): a g16_multi owns one sharded handle per
 * entry of `devices` (an ordinal may repeat: several shards on one GPU) and g16_multi_prove runs the sharded
 * pipeline above on one host thread per shard, moving the slices between devices with peer copies and adding the
 * partial sums on the host.  Same results and error texts as g16_prove.  opts: device / shard_* are ignored. */
typedef struct g16_multi g16_multi;
int g16_multi_create(const uint8_t* zkey, size_t zkey_len, const int32_t* devices, uint32_t ndev,
                     const g16_opts* opts, g16_multi** out);
int g16_multi_prove(g16_multi* m, const uint8_t* wtns, size_t wtns_len, const uint8_t r[32], const uint8_t s[32],
                    g16_proof* out, uint8_t* pub);
int g16_multi_get_info(const g16_multi* m, g16_info* out, uint32_t* n_shards);
void g16_multi_destroy(g16_multi* m);

/* Host-only assembly from gathered partials (no GPU handle needed; reads only the zkey header),
 * and the shard -> point-range map used by g16_create. */
int g16_finish_host(const uint8_t* zkey, size_t zkey_len, const uint8_t* partials, uint32_t count,
                    const uint8_t r[32], const uint8_t s[32], g16_proof* out);
void g16_shard_range(uint32_t total, int32_t rank, int32_t count, uint32_t* lo, uint32_t* hi);

int g16_get_info(const g16_prover* p, g16_info* out);
/* Device-side phase times of the last proof this handle completed (of the last one collected, after g16_prove_batch).
 * Read from the proof's HIP events when called -- a proof itself no longer pays for them -- so call it before the next
 * proof is started on the handle; serialised with the handle's other entry points. */
int g16_get_timings(const g16_prover* p, g16_timings* out);
void g16_destroy(g16_prover* p);
const char* g16_last_error(void);

/* Operator-level entry points, the device twins of ffjavascript's public curve API [EXT]:
 *   g16_fr_fft / g16_fr_ifft          Fr.fft / Fr.ifft   (n Montgomery residues, natural order,
 *                                     in place; ifft includes 1/n)
 *   g16_g1_multiexp / g16_g2_multiexp G1/G2.multiExpAffine(bases LEM affine, scalars standard LE)
 *                                     -> affine STANDARD-form point (64 / 128 bytes)
 *   g16_fr_batch_mul                  Montgomery products out[i] = a[i]*b[i]  (frm_mul)        */
int g16_fr_fft(int device, uint8_t* buf, size_t n);
int g16_fr_ifft(int device, uint8_t* buf, size_t n);
int g16_fr_batch_mul(int device, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int field /*0 Fr, 1 Fq*/);
/* Layer tests: elementwise field op (op: 0 mul, 1 add, 2 sub, 3 toMontgomery, 4 fromMontgomery,
 * 5 neg; field 0 = Fr, 1 = Fq; raw 32-byte images in and out) and elementwise affine point addition
 * a[i]+b[i] on the device (curve 1 = G1, 2 = G2; LEM affine in, STANDARD affine out). */
int g16_field_op(int device, int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
int g16_ec_add(int device, int curve, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n);
/* Layer tests of the arithmetic the kernels actually run (fq29.cuh / fr29.cuh / ec29.cuh: 9 x 29-bit limbs,
 * lazy reduction, Montgomery radix 2^261), the device twins of wasmcurves' f1m_mul / f1m_square / curve add [EXT].
 * Elements are RAW limb images: 9 little-endian u32 limbs + 1 pad word = 40 bytes, value = sum l[i] 2^(29 i)
 * (limbs 0..7 below 2^29; the value may exceed p: the format is lazy), so tests can place inputs at the bounds
 * the kernels rely on.  field 0 = Fr, 1 = Fq.  op: 0 mul(a,b)  1 sqr(a)  2 a*b + c*d (one reduction)
 * 3 a^2 + c*d  4 a+b  5 a+2p-b  6 a+6p-b  7 a+4p-b-2c  8 2p-a  9 mul (row-wise variant)  10 weak_reduce (Fr)
 * 11 zero tests: limb 0 = (a == 0 mod p), limb 1 = the low-limb filter f29_maybe_zero<7>.  b/c/d may be NULL
 * when the op ignores them. */
int g16_f29_op(int device, int field, int op, const uint8_t* a, const uint8_t* b, const uint8_t* c,
               const uint8_t* d, uint8_t* out, size_t n);
/* XYZZ point formulas over that field.  curve 1 = G1 (coordinate = 40 B), 2 = G2 (80 B: a | b of a + b u).
 * acc / out: n XYZZ images (x | y | zz | zzz); q: n affine images (x | y) for op 0, 1, 4, n XYZZ images for op 2.
 * op 0 = x29_madd_fast, the exception-free hot-loop formula (flags[i] = 1 when it asks for the redo pass),
 * 1 = x29_madd (complete), 2 = x29_add, 3 = x29_dbl, 4 = x29_madd after a pack/unpack round trip of q (the
 * resident 64/128-byte base format). */
int g16_x29_op(int device, int curve, int op, const uint8_t* acc, const uint8_t* q, uint8_t* out, uint8_t* flags,
               size_t n);
/* [EXT] snarkjs groth16_prove.js buildABC1 on a staged witness: A_T, B_T, C_T = A_T o B_T as canonical
 * Montgomery(2^256) residues, domain_size * 32 bytes each. */
int g16_qap_eval(g16_prover* p, uint32_t slot, uint8_t* a, uint8_t* b, uint8_t* c);
int g16_g1_multiexp(int device, const uint8_t* bases, const uint8_t* scalars, size_t n,
                    int window_bits, uint8_t out[64]);
int g16_g2_multiexp(int device, const uint8_t* bases, const uint8_t* scalars, size_t n,
                    int window_bits, uint8_t out[128]);

/* Test-only trapdoor setup over a shape-matched synthetic R1CS (the real nzcp_live
 * R1CS cannot be produced offline).  Host-only (no GPU needed).  Emits a snarkjs-layout .zkey and
 * .wtns plus the verification key points (alpha1 | beta2 | gamma2 | delta2 | IC[0..p], affine
 * Montgomery LE, 64/128 bytes each).  Buffers are malloc'd; release with g16_free.
 * Generator spec: oracle/synth.py docstring (both sides follow the same draw order). */
int g16_synth_setup(uint32_t n_vars, uint32_t n_public, uint32_t n_constraints, uint64_t seed,
                    int threads, uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                    uint8_t** vkey, size_t* vkey_len);
/* A fresh satisfying witness for the same synthetic circuit (batch mode, BASELINE config 3): the
 * free wires are redrawn from `wseed`, the slack wires re-solved; the circuit stays that of `seed`.
 * g16_synth_setup uses wseed = seed.  Returns a .wtns image. */
int g16_synth_witness(uint32_t n_vars, uint32_t n_public, uint32_t n_constraints, uint64_t seed,
                      uint64_t wseed, uint8_t** wtns, size_t* wtns_len);
/* Test-only trapdoor setup of a REAL circuit: iden3 .r1cs v1 in (what `circom --r1cs`
 * writes; nPublic = nPubOut + nPubIn), snarkjs-layout .zkey and the verification-key points out (same
 * layout as g16_synth_setup).  Trapdoor from `seed`.  Lets anyone with circom fabricate a key for the
 * real nzcp_live R1CS and benchmark its true shape; NOT a ceremony -- the trapdoor is known. */
int g16_r1cs_setup(const uint8_t* r1cs, size_t r1cs_len, uint64_t seed, int threads, uint8_t** zkey,
                   size_t* zkey_len, uint8_t** vkey, size_t* vkey_len);
/* Where the fixed-base multiplications of every *_setup entry point run ("trapdoor setup at scale on
 * GPU (fixed-base MSM kernel)"; stands in for `snarkjs groth16 setup`, /root/reference/Makefile:30-31 records the PLONK
 * twin): device >= 0 = that HIP device (csrc/setup_gpu.hip: one lane per scalar over a small L2-resident window table,
 * batched conversion to affine), -1 = host threads (the default).  Process-wide; the zkey bytes are the same either way.
 * A setup with the device path selected and no GPU present returns G16_E_NOGPU. */
int g16_setup_device(int device);

/* Test-only: a REAL constraint system for BASELINE config 5 -- `blocks` chained SHA-256 compressions
 * d_{i+1} = SHA-256(d_i) over a 32-byte private message, bit-level R1CS in the style of the circomlib sha256
 * gadgets nzcptpl.circom includes (/root/reference/circuits/nzcptpl.circom:3-6 include list), ~27 k
 * constraints per block; public signals = the 256 digest bits, MSB-first (the NZCP circuit's bit order,
 * /root/reference/test/nzcp.js:41-47).  Trapdoor zkey as g16_synth_setup; r1cs = iden3 .r1cs v1 image.
 * Any output pair may be NULL. */
int g16_sha256_chain_setup(uint32_t blocks, const uint8_t msg[32], uint64_t seed, int threads,
                           uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                           uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len);

/* Test-only: plain SHA-256 of a `len`-byte private message (len fixed at circuit-build time, FIPS 180-4
 * padding, multi-block), same gadgets and wire layout: public signals = the 256 digest bits MSB-first, i.e.
 * the `toBeSignedSha256` slice of the NZCP circuit's public.json when msg = the pass's ToBeSigned
 * (/root/reference/circuits/nzcptpl.circom:447-500, /root/reference/test/nzcp.js:41-47). */
int g16_sha256_message_setup(const uint8_t* msg, uint32_t len, uint64_t seed, int threads,
                             uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                             uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len);

/* Test-only: the NZCP circuit's public interface on a FIXED pass layout -- public signals [0..255] =
 * SHA-256("given,family,dob") bits, [256..511] = SHA-256(ToBeSigned) bits, [512] = exp, exactly the layout and
 * bit order of /root/reference/test/nzcp.js:41-47 -- with the byte offsets of the three strings and of the
 * 4 exp bytes inside ToBeSigned given as circuit constants (the reference finds them by in-circuit CBOR parsing,
 * /root/reference/circuits/cbortpl.circom; not restated).  The credential string reuses the ToBeSigned bit
 * wires, so the three outputs are bound to one ToBeSigned. */
int g16_nzcp_fixed_layout_setup(const uint8_t* tbs, uint32_t len, const uint32_t seg_off[3], const uint32_t seg_len[3],
                                uint32_t exp_off, uint64_t seed, int threads,
                                uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                                uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len);
/* Test-only: the NZCP circuit library as native R1CS gadgets (csrc/nzcp_gadgets.h), the twins of the reference's
 * per-template test circuits /root/reference/circuits/ *_test.circom.  Builds ONE template over the given private
 * inputs, checks every emitted row on the computed witness and returns the template's output signals (in the
 * order the reference declares them).  name: getType getX quinSelector getV decodeUint23 decodeUint readType
 * skipValueScalar skipValue stringEquals readStringLength readMapLength copyString findVCAndExp findCredSubj
 * readCredSubj concatCredSubj sha256Var; params / inputs: the template parameters / input signals in declaration
 * order (arrays flattened).  G16_E_STATE with "constraint not satisfied: ..." where circom's witness generator
 * would throw.  *nout: capacity in, count out. */
int g16_nzcp_gadget(const char* name, const uint32_t* params, uint32_t nparams, const uint64_t* inputs, uint32_t nin,
                    uint64_t* outputs, uint32_t* nout, uint32_t* n_constraints);
/* Test-only: NZCPPubIdentity(params[0..6]) (/root/reference/circuits/nzcptpl.circom:433) with the CBOR search in
 * the circuit -- nzcp_exampleTest.circom = {0, 314, 0, 4, 2, 4, 5}, nzcp_liveTest.circom = {1, 355, 0, 4, 2, 4, 6} --
 * built natively for the given ToBeSigned bytes: R1CS, witness, trapdoor key.  Public signals as
 * /root/reference/test/nzcp.js:41-47.  Any output pair may be NULL. */
int g16_nzcp_circuit_setup(const uint32_t params[7], const uint8_t* tbs, uint32_t len, uint64_t seed, int threads,
                           uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len, uint8_t** vkey,
                           size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len, uint32_t* n_constraints);
void g16_free(void* p);

/* ---------------------------------------------------------------------------------------------------------------
 * Groth16 batch verifier on the device (the acceptance check of SURVEY 3.4).
 * [EXT] snarkjs groth16_verify.js `groth16.verify(vk, publicSignals, proof)`: cpub = IC[0] + sum pub_i IC[i+1], then
 * curve.pairingEq(-pi_a, pi_b, cpub, vk_gamma_2, pi_c, vk_delta_2, vk_alpha_1, vk_beta_2)  (pins
 * /root/reference/yarn.lock:987-1001; verification_key.json layout SURVEY App. A.5).  One handle = one verification
 * key resident on one GPU; a batch call returns one verdict per proof (the verdicts of one-at-a-time verification).
 *
 * vkey: alpha1 (64 B) | beta2 (128) | gamma2 (128) | delta2 (128) | IC[0..n_public] (64 each), affine little-endian,
 *       G2 as x.c0 | x.c1 | y.c0 | y.c1; montgomery = 1: Montgomery residues (what the *_setup entry points return),
 *       0: standard integers (what verification_key.json holds).  Rejected with G16_E_FORMAT: wrong length, a
 *       coordinate >= q, a point off its curve.
 * proofs: g16_proof as g16_prove writes them (standard form).  pubs: count * n_public * 32 bytes, standard LE
 *       (any 256-bit value: snarkjs reduces modulo r, so does the scalar multiplication here).
 * ok[i] = 1 accepted, 0 rejected (pairing product != 1, or a proof coordinate >= q, or a proof point off its curve).
 * No CPU path: G16_E_NOGPU without a HIP device. */
typedef struct g16_verifier g16_verifier;
int g16_verifier_create(const uint8_t* vkey, size_t vkey_len, uint32_t n_public, int montgomery, int device,
                        g16_verifier** out);
int g16_verify_batch(g16_verifier* v, const g16_proof* proofs, const uint8_t* pubs, size_t count, uint8_t* ok);
/* device time of the last batch, ms: [0] vk_x (public-signal MSMs), [1] Miller loops, [2] final exponentiations */
int g16_verifier_timings(const g16_verifier* v, float ms[3]);
void g16_verifier_destroy(g16_verifier* v);
/* Layer-test operator: `count` pairs (P in G1, Q in G2; six standard-form 32-byte words each: px py qx.c0 qx.c1 qy.c0
 * qy.c1, both on their curves, neither infinity) -> twelve standard-form words per pair: the pairing value the verifier
 * kernels compute, as the coefficients (c0, c1) of W^0..W^5 in the tower Fq12 = Fq2[W]/(W^6 - (9+u)). */
int g16_pairing_op(int device, const uint8_t* in, uint32_t count, uint8_t* out);

/* ---------------------------------------------------------------------------------------------------------------
 * PLONK prover on the device.  [EXT] snarkjs 0.4.12 plonk_prove.js `plonk.prove(zkey, wtns)` -- the
 * protocol whose setup the reference scripts (/root/reference/Makefile:30-33: `snarkjs plonk setup`, `zkey export
 * verificationkey`, `zkey export solidityverifier`).  zkey: a PLONK .zkey as `snarkjs plonk setup` writes it (protocol
 * id 2; the Lagrange section 13 is not read).  Errors carry snarkjs's texts: "zkey file is not plonk" (snarkjs's own
 * text says "groth16" there), "Invalid witness length. Circuit: N, witness: M, A", "Copy constraints does not match",
 * "T Polynomial is not divisible", "Polinomial does not divide".
 * blinding: nine 32-byte LE scalars b1..b9 below r (snarkjs: Fr.random() each), or NULL for the OS CSPRNG -- with the
 * same nine scalars the proof is reproducible byte for byte.
 * proof: points affine in standard form (x | y, infinity = zeros), evaluations as standard-form integers: the fields of
 * snarkjs's proof.json in its key order.  pub receives n_public * 32 bytes.  No CPU path (G16_E_NOGPU). */
typedef struct g16_plonk g16_plonk;
typedef struct g16_plonk_proof {
  uint8_t A[64], B[64], C[64], Z[64], T1[64], T2[64], T3[64];
  uint8_t eval_a[32], eval_b[32], eval_c[32], eval_s1[32], eval_s2[32], eval_zw[32], eval_r[32];
  uint8_t Wxi[64], Wxiw[64];
} g16_plonk_proof;
int g16_plonk_create(const uint8_t* zkey, size_t zkey_len, int device, g16_plonk** out);
int g16_plonk_prove(g16_plonk* p, const uint8_t* wtns, size_t wtns_len, const uint8_t* blinding /* 9 * 32 or NULL */,
                    g16_plonk_proof* out, uint8_t* pub);
/* info: nVars (with the addition signals), nPublic, domainSize, nAdditions, nConstraints, addition dependency levels */
int g16_plonk_get_info(const g16_plonk* p, uint32_t info[6]);
/* host wall time of the last proof, ms: [0] witness + round 1, [1] round 2, [2] round 3, [3] round 4, [4] round 5, [5] total */
int g16_plonk_timings(const g16_plonk* p, float ms[6]);
void g16_plonk_destroy(g16_plonk* p);
/* Test-only stand-in for `snarkjs plonk setup c.r1cs pot.ptau c.zkey` (/root/reference/Makefile:31) with a KNOWN tau
 * (derived from seed): iden3 .r1cs v1 in, snarkjs-layout PLONK .zkey out (g16_free).  R1CS -> gates as plonk_setup.js
 * does it; the selector / sigma transforms run on `device` (required), the N + 6 powers of tau through the fixed-base
 * kernel.  with_lagrange = 0 writes an EMPTY section 13 (a prover that derives the public-input polynomial by NTT does
 * not read it; snarkjs does). */
int g16_plonk_setup(const uint8_t* r1cs, size_t r1cs_len, uint64_t seed, int device, int with_lagrange, uint8_t** zkey,
                    size_t* zkey_len);
/* `snarkjs plonk setup c.r1cs pot.ptau c.zkey` itself (/root/reference/Makefile:31: powersOfTau28_hez_final_22.ptau):
 * the powers come from a .ptau v1 image ([EXT] snarkjs powersoftau_utils.js; sections 1-3 are read), the eight
 * selector / sigma commitments are MSMs over them on the device.  "circuit too big for this power of tau ceremony.
 * G > 2**P" when the gates do not fit. */
int g16_plonk_setup_ptau(const uint8_t* r1cs, size_t r1cs_len, const uint8_t* ptau, size_t ptau_len, int device,
                         int with_lagrange, uint8_t** zkey, size_t* zkey_len);
/* the same from / to files (inputs mapped read-only): for hosts whose buffers end at 2 GB (Node.js) -- a 2^22 ceremony
 * file is 4.6 GB, the key it yields 5.8 GB without the Lagrange section */
int g16_plonk_setup_files(const char* r1cs_path, const char* ptau_path, const char* zkey_path, int device, int with_lagrange);

/* PLONK batch verifier on the device: [EXT] snarkjs 0.4.12 plonk_verify.js `plonk.verify(vk, publicSignals, proof)` for
 * many proofs against one key, one verdict per proof (transcript and scalar arithmetic on host threads, the twenty
 * scalar multiplications, two Miller loops and the final exponentiation of every proof on the device).
 * vkey (712 bytes, what verification_key.json holds, standard-form little-endian): power u32 | nPublic u32 | k1 (32) |
 * k2 (32) | Qm Ql Qr Qo Qc S1 S2 S3 (64 each: x | y) | X_2 (128: x.c0 | x.c1 | y.c0 | y.c1).
 * proofs: g16_plonk_proof as g16_plonk_prove writes them; pubs: count * nPublic * 32 bytes.  ok[i] = 1 / 0. */
typedef struct g16_plonk_verifier g16_plonk_verifier;
int g16_plonk_verifier_create(const uint8_t* vkey, size_t vkey_len, int device, g16_plonk_verifier** out);
int g16_plonk_verify_batch(g16_plonk_verifier* v, const g16_plonk_proof* proofs, const uint8_t* pubs, size_t count, uint8_t* ok);
void g16_plonk_verifier_destroy(g16_plonk_verifier* v);

#ifdef __cplusplus
}
#endif
#endif /* G16_PROVER_H */
