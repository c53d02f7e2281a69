"""nzcp-circom_amd -- MI355X-native Groth16 prover for the compiled NZCP circuits.

This Python module is PLUMBING: a ctypes binding over the C ABI of libg16hip.so
(include/g16_prover.h) used by tests/ and bench.py.  The product host is the Node.js shim in
nzcp-circom_amd/js (snarkjs-shaped `groth16.prove`); both sit on the same C ABI.

There is no CPU fallback: `load()` raises if the shared library is missing, and every compute
entry point returns G16_E_NOGPU (raised here as G16Error) when no HIP device is present.
The API mirrors snarkjs 0.4.12 `groth16.prove` (pin /root/reference/yarn.lock:987-1001):
`Prover(zkey).prove(wtns, r, s) -> (proof_dict, public_signals)` with snarkjs's JSON shapes and
error messages.
"""
import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("G16_LIB_NAME", "libg16hip.so"))   # (G16_LIB_NAME: A/B of two builds)
PARTIAL_BYTES = 128 * 4 + 256
LAZY_FR_BYTES = 40

_lib = None


class G16Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("window_bits", C.c_int32), ("task_len", C.c_int32), ("flags", C.c_uint32)]


class Proof(C.Structure):
    _fields_ = [("a", C.c_uint8 * 64), ("b", C.c_uint8 * 128), ("c", C.c_uint8 * 64)]


class PlonkProof(C.Structure):
    _fields_ = ([(k, C.c_uint8 * 64) for k in ("A", "B", "C", "Z", "T1", "T2", "T3")] +
                [(k, C.c_uint8 * 32) for k in ("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw", "eval_r")] +
                [("Wxi", C.c_uint8 * 64), ("Wxiw", C.c_uint8 * 64)])


class Info(C.Structure):
    _fields_ = [("n_vars", C.c_uint32), ("n_public", C.c_uint32), ("domain_size", C.c_uint32),
                ("n_coefs", C.c_uint32), ("n_a", C.c_uint32), ("n_b1", C.c_uint32),
                ("n_b2", C.c_uint32), ("n_c", C.c_uint32), ("n_h", C.c_uint32),
                ("window_bits", C.c_uint32 * 5)]


class Timings(C.Structure):
    _fields_ = [("upload_ms", C.c_float), ("qap_ms", C.c_float), ("ntt_ms", C.c_float),
                ("msm_ms", C.c_float * 5), ("tail_ms", C.c_float), ("total_ms", C.c_float),
                ("msm_accum_kernel_ms", C.c_float * 5)]


EXPORTS = ["g16_create", "g16_prove", "g16_prove_batch", "g16_stage_witness", "g16_prove_staged",
           "g16_prove_partial", "g16_prove_finish", "g16_get_info", "g16_get_timings", "g16_destroy",
           "g16_last_error", "g16_fr_fft", "g16_fr_ifft", "g16_fr_batch_mul", "g16_field_op",
           "g16_ec_add", "g16_g1_multiexp", "g16_g2_multiexp", "g16_synth_setup",
           "g16_synth_witness", "g16_free", "g16_finish_host", "g16_shard_range", "g16_r1cs_setup",
           "g16_sha256_chain_setup", "g16_sha256_message_setup", "g16_nzcp_fixed_layout_setup",
           "g16_f29_op", "g16_x29_op", "g16_qap_eval", "g16_shard_begin", "g16_shard_end",
           "g16_multi_create", "g16_multi_prove", "g16_multi_get_info", "g16_multi_destroy",
           "g16_nzcp_gadget", "g16_nzcp_circuit_setup", "g16_setup_device",
           "g16_verifier_create", "g16_verify_batch", "g16_verifier_timings", "g16_verifier_destroy", "g16_pairing_op",
           "g16_plonk_create", "g16_plonk_prove", "g16_plonk_get_info", "g16_plonk_destroy", "g16_plonk_setup", "g16_plonk_timings", "g16_plonk_setup_ptau", "g16_plonk_setup_files",
           "g16_plonk_verifier_create", "g16_plonk_verify_batch", "g16_plonk_verifier_destroy"]


def load():
    """dlopen libg16hip.so; fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise G16Error(-4, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(LIB_PATH)
    u8p, sz, vp = C.POINTER(C.c_uint8), C.c_size_t, C.c_void_p
    lib.g16_last_error.restype = C.c_char_p
    lib.g16_create.argtypes = [C.c_char_p, sz, C.POINTER(Opts), C.POINTER(vp)]
    lib.g16_prove.argtypes = [vp, C.c_char_p, sz, C.c_char_p, C.c_char_p, C.POINTER(Proof), C.c_char_p]
    lib.g16_stage_witness.argtypes = [vp, C.c_uint32, C.c_char_p, sz]
    lib.g16_prove_staged.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_char_p, C.POINTER(Proof), C.c_char_p]
    lib.g16_prove_partial.argtypes = [vp, C.c_uint32, C.c_char_p]
    lib.g16_prove_finish.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p,
                                     C.POINTER(Proof), C.c_char_p]
    lib.g16_prove_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(sz), sz, C.c_char_p,
                                    C.POINTER(Proof), C.c_char_p]
    lib.g16_get_info.argtypes = [vp, C.POINTER(Info)]
    lib.g16_get_timings.argtypes = [vp, C.POINTER(Timings)]
    lib.g16_destroy.argtypes = [vp]
    lib.g16_destroy.restype = None
    lib.g16_fr_fft.argtypes = [C.c_int, C.c_char_p, sz]
    lib.g16_fr_ifft.argtypes = [C.c_int, C.c_char_p, sz]
    lib.g16_fr_batch_mul.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, sz, C.c_int]
    lib.g16_field_op.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, sz]
    lib.g16_ec_add.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, sz]
    lib.g16_f29_op.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz]
    lib.g16_x29_op.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz]
    lib.g16_qap_eval.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.g16_shard_begin.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    lib.g16_shard_end.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.c_char_p]
    lib.g16_nzcp_gadget.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.g16_nzcp_circuit_setup.argtypes = [C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32, C.c_uint64, C.c_int] + \
        [C.c_void_p] * 8 + [C.POINTER(C.c_uint32)]
    lib.g16_multi_create.argtypes = [C.c_char_p, sz, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(Opts), C.POINTER(vp)]
    lib.g16_multi_prove.argtypes = [vp, C.c_char_p, sz, C.c_char_p, C.c_char_p, C.POINTER(Proof), C.c_char_p]
    lib.g16_multi_get_info.argtypes = [vp, C.POINTER(Info), C.POINTER(C.c_uint32)]
    lib.g16_multi_destroy.argtypes = [vp]
    lib.g16_multi_destroy.restype = None
    lib.g16_g1_multiexp.argtypes = [C.c_int, C.c_char_p, C.c_char_p, sz, C.c_int, C.c_char_p]
    lib.g16_g2_multiexp.argtypes = [C.c_int, C.c_char_p, C.c_char_p, sz, C.c_int, C.c_char_p]
    lib.g16_synth_setup.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int,
                                    C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz),
                                    C.POINTER(vp), C.POINTER(sz)]
    lib.g16_synth_witness.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                      C.POINTER(vp), C.POINTER(sz)]
    lib.g16_setup_device.argtypes = [C.c_int]
    lib.g16_verifier_create.argtypes = [C.c_char_p, sz, C.c_uint32, C.c_int, C.c_int, C.POINTER(vp)]
    lib.g16_verify_batch.argtypes = [vp, C.c_char_p, C.c_char_p, sz, C.c_char_p]
    lib.g16_verifier_timings.argtypes = [vp, C.POINTER(C.c_float)]
    lib.g16_verifier_destroy.argtypes = [vp]
    lib.g16_verifier_destroy.restype = None
    lib.g16_pairing_op.argtypes = [C.c_int, C.c_char_p, C.c_uint32, C.c_char_p]
    lib.g16_plonk_create.argtypes = [C.c_char_p, sz, C.c_int, C.POINTER(vp)]
    lib.g16_plonk_prove.argtypes = [vp, C.c_char_p, sz, C.c_char_p, C.POINTER(PlonkProof), C.c_char_p]
    lib.g16_plonk_get_info.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.g16_plonk_destroy.argtypes = [vp]
    lib.g16_plonk_destroy.restype = None
    lib.g16_plonk_timings.argtypes = [vp, C.POINTER(C.c_float)]
    lib.g16_plonk_setup.argtypes = [C.c_char_p, sz, C.c_uint64, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(sz)]
    lib.g16_plonk_verifier_create.argtypes = [C.c_char_p, sz, C.c_int, C.POINTER(vp)]
    lib.g16_plonk_verify_batch.argtypes = [vp, C.c_char_p, C.c_char_p, sz, C.c_char_p]
    lib.g16_plonk_verifier_destroy.argtypes = [vp]
    lib.g16_plonk_verifier_destroy.restype = None
    lib.g16_plonk_setup_files.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    lib.g16_plonk_setup_ptau.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(sz)]
    lib.g16_r1cs_setup.argtypes = [C.c_char_p, sz, C.c_uint64, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz)]
    lib.g16_sha256_chain_setup.argtypes = [C.c_uint32, C.c_char_p, C.c_uint64, C.c_int] + [C.c_void_p] * 8
    lib.g16_sha256_message_setup.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_int] + [C.c_void_p] * 8
    lib.g16_nzcp_fixed_layout_setup.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                                C.c_uint32, C.c_uint64, C.c_int] + [C.c_void_p] * 8
    lib.g16_finish_host.argtypes = [C.c_char_p, sz, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p, C.POINTER(Proof)]
    lib.g16_shard_range.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.g16_shard_range.restype = None
    lib.g16_free.argtypes = [vp]
    lib.g16_free.restype = None
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise G16Error(rc, load().g16_last_error().decode("utf-8", "replace"))


def _take(ptr, n):
    """Copy a malloc'd buffer into bytes and free it."""
    # (string_at takes a C int size: a 2^22-constraint zkey is > 2 GiB)
    data = bytes((C.c_char * n.value).from_address(ptr.value)) if n.value else b""
    load().g16_free(ptr)
    return data


# ------------------------------------------------------------------ snarkjs-shaped helpers
def _dec(b):
    return str(int.from_bytes(b, "little"))


def proof_to_obj(pr):
    """g16_proof -> the object snarkjs returns (decimal strings, key order of groth16_prove.js)."""
    a, b, c = bytes(pr.a), bytes(pr.b), bytes(pr.c)

    def g1(x):
        return ["0", "1", "0"] if x == bytes(64) else [_dec(x[:32]), _dec(x[32:]), "1"]
    if b == bytes(128):
        pb = [["0", "0"], ["1", "0"], ["0", "0"]]
    else:
        pb = [[_dec(b[0:32]), _dec(b[32:64])], [_dec(b[64:96]), _dec(b[96:128])], ["1", "0"]]
    return {"pi_a": g1(a), "pi_b": pb, "pi_c": g1(c), "protocol": "groth16", "curve": "bn128"}


def stringify(obj):
    """JSON.stringify(obj, null, 1), what the snarkjs CLI writes to proof.json / public.json."""
    return json.dumps(obj, indent=1, separators=(",", ": "))


class Prover:
    """Resident proving key on one GPU (or one shard of it)."""

    def __init__(self, zkey, device=0, shard_rank=0, shard_count=1, window_bits=0, task_len=0, precomp=0):
        lib = load()
        if isinstance(zkey, (str, os.PathLike)):
            with open(zkey, "rb") as f:
                zkey = f.read()
        self._h = C.c_void_p()
        opts = Opts(device, shard_rank, shard_count, window_bits, task_len, (precomp & 0xff) << 8)
        _check(lib.g16_create(zkey, len(zkey), C.byref(opts), C.byref(self._h)))
        self.info = Info()
        _check(lib.g16_get_info(self._h, C.byref(self.info)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().g16_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _pub(self):
        return C.create_string_buffer(max(1, self.info.n_public * 32))

    def _unpack(self, pr, pub):
        p = self.info.n_public
        return proof_to_obj(pr), [_dec(pub.raw[i * 32:(i + 1) * 32]) for i in range(p)]

    def prove(self, wtns, r=None, s=None):
        """snarkjs groth16.prove(zkey, wtns) -> (proof, publicSignals); r, s = 32-byte LE or None."""
        if isinstance(wtns, (str, os.PathLike)):
            with open(wtns, "rb") as f:
                wtns = f.read()
        pr, pub = Proof(), self._pub()
        _check(load().g16_prove(self._h, wtns, len(wtns), r, s, C.byref(pr), pub))
        return self._unpack(pr, pub)

    def stage(self, slot, wtns):
        _check(load().g16_stage_witness(self._h, slot, wtns, len(wtns)))

    def prove_staged(self, slot, r=None, s=None):
        pr, pub = Proof(), self._pub()
        _check(load().g16_prove_staged(self._h, slot, r, s, C.byref(pr), pub))
        return self._unpack(pr, pub)

    def prove_staged_raw(self, slot, r, s, pr, pub):
        """Timed path of bench.py: no Python-side formatting."""
        return load().g16_prove_staged(self._h, slot, r, s, C.byref(pr), pub)

    def prove_partial(self, slot):
        buf = C.create_string_buffer(PARTIAL_BYTES)
        _check(load().g16_prove_partial(self._h, slot, buf))
        return buf.raw

    def prove_finish(self, slot, partials, r=None, s=None):
        pr, pub = Proof(), self._pub()
        blob = b"".join(partials)
        _check(load().g16_prove_finish(self._h, slot, blob, len(partials), r, s, C.byref(pr), pub))
        return self._unpack(pr, pub)

    def shard_begin(self, slot, vec_mask, out_ptrs):
        """Sharded H pipeline, first half: out_ptrs = three addresses (ints; 0 where the mask bit is clear) of
        buffers of domain_size * LAZY_FR_BYTES bytes the device can write (host or device memory)."""
        arr = (C.c_void_p * 3)(*[C.c_void_p(p or None) for p in out_ptrs])
        _check(load().g16_shard_begin(self._h, slot, vec_mask, arr))

    def shard_end(self, slot, slice_ptrs):
        """... second half: the three slices [lo, hi) of this shard's H range -> the partial-sum blob."""
        arr = (C.c_void_p * 3)(*[C.c_void_p(p or None) for p in slice_ptrs])
        buf = C.create_string_buffer(PARTIAL_BYTES)
        _check(load().g16_shard_end(self._h, slot, arr, buf))
        return buf.raw

    def qap_eval(self, slot):
        """buildABC1 of a staged witness -> (A_T, B_T, C_T) as lists of canonical Montgomery(2^256) residues."""
        n = self.info.domain_size
        bufs = [C.create_string_buffer(n * 32) for _ in range(3)]
        _check(load().g16_qap_eval(self._h, slot, *bufs))
        return [[int.from_bytes(b.raw[i * 32:(i + 1) * 32], "little") for i in range(n)] for b in bufs]

    def timings(self):
        t = Timings()
        _check(load().g16_get_timings(self._h, C.byref(t)))
        return {"upload_ms": t.upload_ms, "qap_ms": t.qap_ms, "ntt_ms": t.ntt_ms,
                "msm_ms": list(t.msm_ms), "tail_ms": t.tail_ms, "total_ms": t.total_ms,
                "msm_accum_kernel_ms": list(t.msm_accum_kernel_ms)}


class MultiProver:
    """One process, several GPUs: one shard of the key per entry of `devices` (g16_multi_*)."""

    def __init__(self, zkey, devices, window_bits=0, task_len=0):
        lib = load()
        self._h = C.c_void_p()
        opts = Opts(0, 0, 1, window_bits, task_len, 0)
        devs = (C.c_int32 * len(devices))(*devices)
        _check(lib.g16_multi_create(zkey, len(zkey), devs, len(devices), C.byref(opts), C.byref(self._h)))
        self.info = Info()
        ns = C.c_uint32()
        _check(lib.g16_multi_get_info(self._h, C.byref(self.info), C.byref(ns)))
        self.n_shards = ns.value

    def prove(self, wtns, r=None, s=None):
        pr, pub = Proof(), C.create_string_buffer(max(1, self.info.n_public * 32))
        _check(load().g16_multi_prove(self._h, wtns, len(wtns), r, s, C.byref(pr), pub))
        p = self.info.n_public
        return proof_to_obj(pr), [_dec(pub.raw[i * 32:(i + 1) * 32]) for i in range(p)]

    def prove_raw(self, wtns, r, s, pr, pub):
        return load().g16_multi_prove(self._h, wtns, len(wtns), r, s, C.byref(pr), pub)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().g16_multi_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def finish_host(zkey, partials, r, s):
    """Assemble a proof from gathered partial-sum blobs on the host (no GPU handle)."""
    pr = Proof()
    blob = b"".join(partials)
    _check(load().g16_finish_host(zkey, len(zkey), blob, len(partials), r, s, C.byref(pr)))
    return proof_to_obj(pr)


def shard_vector_owner(v, count):
    """Sharded H pipeline: the shard that evaluates vector v (0 = A, 1 = B, 2 = C) on the coset."""
    return v % count


def shard_range(total, rank, count):
    lo, hi = C.c_uint32(), C.c_uint32()
    load().g16_shard_range(total, rank, count, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


# ------------------------------------------------------------------ operator-level (ffjavascript twins)
def fr_fft(buf, inverse=False, device=0):
    out = C.create_string_buffer(bytes(buf), len(buf))
    fn = load().g16_fr_ifft if inverse else load().g16_fr_fft
    _check(fn(device, out, len(buf) // 32))
    return out.raw


def field_op(field, op, a, b, device=0):
    out = C.create_string_buffer(len(a))
    _check(load().g16_field_op(device, field, op, a, b, out, len(a) // 32))
    return out.raw


def ec_add(curve, a, b, device=0):
    psz = 128 if curve == 2 else 64
    out = C.create_string_buffer(len(a))
    _check(load().g16_ec_add(device, curve, a, b, out, len(a) // psz))
    return out.raw


# raw limb images of the kernels' lazy 9 x 29-bit format (layer tests of the arithmetic the hot path runs)
F29_BYTES = 40


def f29_pack(v):
    """int (below 2^293) -> raw F29 image: limbs 0..7 of 29 bits, limb 8 takes the rest, one pad word."""
    out = bytearray(F29_BYTES)
    for i in range(8):
        out[4 * i:4 * i + 4] = ((v >> (29 * i)) & 0x1FFFFFFF).to_bytes(4, "little")
    out[32:36] = (v >> 232).to_bytes(4, "little")
    return bytes(out)


def f29_unpack(b):
    return sum(int.from_bytes(b[4 * i:4 * i + 4], "little") << (29 * i) for i in range(9))


def f29_op(field, op, a, b=None, c=None, d=None, device=0):
    """a..d: lists of ints (values of the lazy elements) -> list of ints."""
    n = len(a)
    enc = [None if v is None else b"".join(f29_pack(x) for x in v) for v in (a, b, c, d)]
    out = C.create_string_buffer(n * F29_BYTES)
    _check(load().g16_f29_op(device, field, op, enc[0], enc[1], enc[2], enc[3], out, n))
    return [f29_unpack(out.raw[i * F29_BYTES:(i + 1) * F29_BYTES]) for i in range(n)]


def x29_op(curve, op, acc, q=None, device=0):
    """acc: list of XYZZ coordinate tuples (4 ints for G1, 4 pairs for G2); q: affine (2) or XYZZ (4) tuples.
    -> (list of XYZZ tuples, list of redo flags)."""
    n = len(acc)

    def enc(pts):
        if pts is None:
            return None
        if curve == 1:
            return b"".join(f29_pack(x) for pt in pts for x in pt)
        return b"".join(f29_pack(x) for pt in pts for co in pt for x in co)
    cb = F29_BYTES * (2 if curve == 2 else 1)
    out = C.create_string_buffer(n * 4 * cb)
    flags = C.create_string_buffer(max(1, n))
    _check(load().g16_x29_op(device, curve, op, enc(acc), enc(q), out, flags, n))
    res = []
    for i in range(n):
        co = []
        for k in range(4):
            raw = out.raw[(4 * i + k) * cb:(4 * i + k + 1) * cb]
            co.append(f29_unpack(raw) if curve == 1 else (f29_unpack(raw[:F29_BYTES]), f29_unpack(raw[F29_BYTES:])))
        res.append(tuple(co))
    return res, list(flags.raw[:n])


def multiexp(curve, bases, scalars, window_bits=0, device=0):
    psz = 128 if curve == 2 else 64
    out = C.create_string_buffer(psz)
    fn = load().g16_g2_multiexp if curve == 2 else load().g16_g1_multiexp
    _check(fn(device, bases, scalars, len(scalars) // 32, window_bits, out))
    return out.raw


# ------------------------------------------------------------------ test-only setup tool
def synth_setup(n_vars, n_public, n_constraints, seed, threads=0):
    """-> (zkey bytes, wtns bytes, vkey bytes)  (host only, no GPU needed)."""
    lib = load()
    z, w, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    zl, wl, vl = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _check(lib.g16_synth_setup(n_vars, n_public, n_constraints, seed, threads, C.byref(z), C.byref(zl),
                               C.byref(w), C.byref(wl), C.byref(v), C.byref(vl)))
    return _take(z, zl), _take(w, wl), _take(v, vl)


def sha256_chain_setup(blocks, msg, seed, threads=0, want_zkey=True, want_r1cs=False, chain=True):
    """SHA-256 chain circuit (chain=True: `blocks` compressions over a 32-byte message) or plain SHA-256 of `msg`
    (chain=False): a real constraint system -> dict(zkey, wtns, vkey, r1cs) of bytes / None."""
    lib = load()
    assert not chain or len(msg) == 32
    ptrs = [C.c_void_p() for _ in range(4)]
    lens = [C.c_size_t() for _ in range(4)]
    want = [want_zkey, True, want_zkey, want_r1cs]
    args = []
    for p_, l_, w_ in zip(ptrs, lens, want):
        args += [C.byref(p_) if w_ else None, C.byref(l_) if w_ else None]
    if chain:
        _check(lib.g16_sha256_chain_setup(blocks, msg, seed, threads, *args))
    else:
        _check(lib.g16_sha256_message_setup(msg, len(msg), seed, threads, *args))
    out = [(_take(p_, l_) if w_ else None) for p_, l_, w_ in zip(ptrs, lens, want)]
    return {"zkey": out[0], "wtns": out[1], "vkey": out[2], "r1cs": out[3]}


def nzcp_fixed_layout_setup(tbs, segs, exp_off, seed, threads=0, want_zkey=True, want_r1cs=False):
    """The NZCP public interface on a fixed pass layout: segs = [(offset, length)] x 3 of givenName, familyName,
    dob inside ToBeSigned; exp_off = offset of the 4 big-endian exp bytes.  -> dict(zkey, wtns, vkey, r1cs)."""
    lib = load()
    ptrs = [C.c_void_p() for _ in range(4)]
    lens = [C.c_size_t() for _ in range(4)]
    want = [want_zkey, True, want_zkey, want_r1cs]
    args = []
    for p_, l_, w_ in zip(ptrs, lens, want):
        args += [C.byref(p_) if w_ else None, C.byref(l_) if w_ else None]
    off = (C.c_uint32 * 3)(*[o for o, _ in segs])
    ln = (C.c_uint32 * 3)(*[n for _, n in segs])
    _check(lib.g16_nzcp_fixed_layout_setup(tbs, len(tbs), off, ln, exp_off, seed, threads, *args))
    out = [(_take(p_, l_) if w_ else None) for p_, l_, w_ in zip(ptrs, lens, want)]
    return {"zkey": out[0], "wtns": out[1], "vkey": out[2], "r1cs": out[3]}


def nzcp_gadget(name, params, inputs, max_out=4096):
    """One template of the NZCP circuit library built natively over `inputs` (the reference's *_test.circom
    twins): -> (outputs, n_constraints); raises G16Error where circom's witness generator would throw."""
    lib = load()
    prm = (C.c_uint32 * max(1, len(params)))(*params)
    inp = (C.c_uint64 * max(1, len(inputs)))(*inputs)
    out = (C.c_uint64 * max_out)()
    nout, ncons = C.c_uint32(max_out), C.c_uint32()
    _check(lib.g16_nzcp_gadget(name.encode(), prm, len(params), inp, len(inputs), out, C.byref(nout), C.byref(ncons)))
    return list(out[:nout.value]), ncons.value


NZCP_EXAMPLE_PARAMS = (0, 314, 0, 4, 2, 4, 5)    # /root/reference/circuits/nzcp_exampleTest.circom
NZCP_LIVE_PARAMS = (1, 355, 0, 4, 2, 4, 6)       # /root/reference/circuits/nzcp_liveTest.circom


def nzcp_circuit_setup(params, tbs, seed, threads=0, want_zkey=True, want_r1cs=False):
    """NZCPPubIdentity(*params) with the CBOR search in the circuit, built natively for the ToBeSigned bytes `tbs`
    -> dict(zkey, wtns, vkey, r1cs, n_constraints)."""
    lib = load()
    ptrs = [C.c_void_p() for _ in range(4)]
    lens = [C.c_size_t() for _ in range(4)]
    want = [want_zkey, True, want_zkey, want_r1cs]
    args = []
    for p_, l_, w_ in zip(ptrs, lens, want):
        args += [C.byref(p_) if w_ else None, C.byref(l_) if w_ else None]
    ncons = C.c_uint32()
    prm = (C.c_uint32 * 7)(*params)
    _check(lib.g16_nzcp_circuit_setup(prm, tbs, len(tbs), seed, threads, *args, C.byref(ncons)))
    out = [(_take(p_, l_) if w_ else None) for p_, l_, w_ in zip(ptrs, lens, want)]
    return {"zkey": out[0], "wtns": out[1], "vkey": out[2], "r1cs": out[3], "n_constraints": ncons.value}


def sha256_message_setup(msg, seed, threads=0, want_zkey=True, want_r1cs=False):
    return sha256_chain_setup(0, msg, seed, threads, want_zkey, want_r1cs, chain=False)


def setup_device(device):
    """Where the fixed-base multiplications of the *_setup builders run: a HIP device ordinal, or -1 = host threads."""
    _check(load().g16_setup_device(int(device)))


def r1cs_setup(r1cs, seed, threads=0):
    """Trapdoor setup of a real .r1cs (bytes) -> (zkey bytes, vkey point bytes)."""
    lib = load()
    z, v = C.c_void_p(), C.c_void_p()
    zl, vl = C.c_size_t(), C.c_size_t()
    _check(lib.g16_r1cs_setup(r1cs, len(r1cs), seed, threads, C.byref(z), C.byref(zl), C.byref(v), C.byref(vl)))
    return _take(z, zl), _take(v, vl)


_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def vkey_json(vkey, n_public):
    """vkey point bytes (alpha1 | beta2 | gamma2 | delta2 | IC[], affine Montgomery LE) -> the
    verification_key.json object snarkjs's `groth16.verify` takes (without the redundant vk_alphabeta_12)."""
    rinv = pow(1 << 256, -1, _Q)

    def fq(b):
        return str(int.from_bytes(b, "little") * rinv % _Q)

    def g1(b):
        return ["0", "1", "0"] if b == bytes(64) else [fq(b[:32]), fq(b[32:64]), "1"]

    def g2(b):
        if b == bytes(128):
            return [["0", "0"], ["1", "0"], ["0", "0"]]
        return [[fq(b[0:32]), fq(b[32:64])], [fq(b[64:96]), fq(b[96:128])], ["1", "0"]]
    return {"protocol": "groth16", "curve": "bn128", "nPublic": n_public,
            "vk_alpha_1": g1(vkey[0:64]), "vk_beta_2": g2(vkey[64:192]), "vk_gamma_2": g2(vkey[192:320]),
            "vk_delta_2": g2(vkey[320:448]),
            "IC": [g1(vkey[448 + 64 * i:512 + 64 * i]) for i in range(n_public + 1)]}


# ------------------------------------------------------------------ verifier (snarkjs groth16.verify, batched)
def _le32(x):
    return (int(x) % (1 << 256)).to_bytes(32, "little")


def proof_from_obj(obj):
    """proof.json object (decimal strings) -> g16_proof bytes (affine standard LE; infinity = zeros)."""
    def g1(t):
        return bytes(64) if str(t[2]) == "0" else _le32(t[0]) + _le32(t[1])
    b = obj["pi_b"]
    pb = bytes(128) if (str(b[2][0]), str(b[2][1])) == ("0", "0") else _le32(b[0][0]) + _le32(b[0][1]) + _le32(b[1][0]) + _le32(b[1][1])
    return g1(obj["pi_a"]) + pb + g1(obj["pi_c"])


def vkey_from_json(vk):
    """verification_key.json object -> (point bytes in STANDARD form, nPublic) for Verifier(..., montgomery=False)."""
    def g1(t):
        return bytes(64) if str(t[2]) == "0" else _le32(t[0]) + _le32(t[1])

    def g2(t):
        if (str(t[2][0]), str(t[2][1])) == ("0", "0"):
            return bytes(128)
        return _le32(t[0][0]) + _le32(t[0][1]) + _le32(t[1][0]) + _le32(t[1][1])
    n_public = int(vk["nPublic"])
    if len(vk["IC"]) != n_public + 1:
        raise G16Error(-2, "verification key: IC length does not match nPublic")
    return (g1(vk["vk_alpha_1"]) + g2(vk["vk_beta_2"]) + g2(vk["vk_gamma_2"]) + g2(vk["vk_delta_2"]) +
            b"".join(g1(t) for t in vk["IC"])), n_public


class Verifier:
    """Resident verification key on one GPU: snarkjs `groth16.verify(vk, publicSignals, proof)` for batches."""

    def __init__(self, vkey, n_public=None, montgomery=True, device=0):
        if isinstance(vkey, dict):
            vkey, n_public = vkey_from_json(vkey)
            montgomery = False
        self.n_public = int(n_public)
        self._h = C.c_void_p()
        _check(load().g16_verifier_create(vkey, len(vkey), self.n_public, 1 if montgomery else 0, device, C.byref(self._h)))

    def verify_raw(self, proofs, pubs, count):
        """proofs: count * 256 bytes of g16_proof; pubs: count * n_public * 32 bytes -> list of bools."""
        ok = C.create_string_buffer(max(1, count))
        _check(load().g16_verify_batch(self._h, proofs, pubs, count, ok))
        return [b != 0 for b in ok.raw[:count]]

    def verify(self, public_signals, proof):
        """One proof: snarkjs's signature (publicSignals as decimal strings / ints, proof as the proof.json object)."""
        if len(public_signals) != self.n_public:
            return False
        return self.verify_raw(proof_from_obj(proof), b"".join(_le32(x) for x in public_signals), 1)[0]

    def verify_batch(self, items):
        """items: [(public_signals, proof_obj)] -> list of bools."""
        bad = [len(ps) != self.n_public for ps, _ in items]
        pr = b"".join(proof_from_obj(p) for _, p in items)
        pub = b"".join(b"".join(_le32(x) for x in (ps if not b else [0] * self.n_public)) for (ps, _), b in zip(items, bad))
        res = self.verify_raw(pr, pub, len(items))
        return [r and not b for r, b in zip(res, bad)]

    def timings(self):
        ms = (C.c_float * 3)()
        _check(load().g16_verifier_timings(self._h, ms))
        return list(ms)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().g16_verifier_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


_PLONK_G1 = ("A", "B", "C", "Z", "T1", "T2", "T3")
_PLONK_EV = ("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw", "eval_r")


class PlonkProver:
    """Resident PLONK proving key on one GPU: snarkjs `plonk.prove(zkey, wtns)` -> (proof, publicSignals)."""

    def __init__(self, zkey, device=0):
        if isinstance(zkey, (str, os.PathLike)):
            with open(zkey, "rb") as f:
                zkey = f.read()
        self._h = C.c_void_p()
        _check(load().g16_plonk_create(zkey, len(zkey), device, C.byref(self._h)))
        info = (C.c_uint32 * 6)()
        _check(load().g16_plonk_get_info(self._h, info))
        self.n_vars, self.n_public, self.domain_size, self.n_additions, self.n_constraints, self.levels = list(info)

    def prove_raw(self, wtns, blinding=None):
        pr = PlonkProof()
        pub = C.create_string_buffer(max(1, self.n_public * 32))
        _check(load().g16_plonk_prove(self._h, wtns, len(wtns), blinding, C.byref(pr), pub))
        return pr, pub.raw[:self.n_public * 32]

    def prove(self, wtns, blinding=None):
        """blinding: None, or the nine scalars b1..b9 as ints (reproducible proof)."""
        if blinding is not None and not isinstance(blinding, (bytes, bytearray)):
            blinding = b"".join(int(x).to_bytes(32, "little") for x in blinding)
        pr, pub = self.prove_raw(wtns, blinding)
        return plonk_proof_to_obj(pr), [_dec(pub[i * 32:(i + 1) * 32]) for i in range(self.n_public)]

    def timings(self):
        ms = (C.c_float * 6)()
        _check(load().g16_plonk_timings(self._h, ms))
        return dict(zip(("witness_round1", "round2", "round3", "round4", "round5", "total"), [round(x, 3) for x in ms]))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().g16_plonk_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def plonk_proof_from_obj(o):
    """proof.json object of `plonk prove` -> g16_plonk_proof bytes (standard LE)."""
    def g1(t):
        return bytes(64) if str(t[2]) == "0" else _le32(t[0]) + _le32(t[1])
    return (b"".join(g1(o[k]) for k in _PLONK_G1) + b"".join(_le32(o[k]) for k in _PLONK_EV) + g1(o["Wxi"]) + g1(o["Wxiw"]))


def plonk_vkey_bytes(vk):
    """verification_key.json object of a PLONK key -> the C ABI's 712-byte image."""
    def g1(t):
        return bytes(64) if str(t[2]) == "0" else _le32(t[0]) + _le32(t[1])
    x2 = vk["X_2"]
    return (int(vk["power"]).to_bytes(4, "little") + int(vk["nPublic"]).to_bytes(4, "little") + _le32(vk["k1"]) + _le32(vk["k2"]) +
            b"".join(g1(vk[k]) for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3")) +
            _le32(x2[0][0]) + _le32(x2[0][1]) + _le32(x2[1][0]) + _le32(x2[1][1]))


class PlonkVerifier:
    """snarkjs `plonk.verify(vk, publicSignals, proof)` for batches: one verdict per proof."""

    def __init__(self, vk, device=0):
        self.n_public = int(vk["nPublic"])
        b = plonk_vkey_bytes(vk)
        self._h = C.c_void_p()
        _check(load().g16_plonk_verifier_create(b, len(b), device, C.byref(self._h)))

    def verify_batch(self, items):
        """items: [(public_signals, proof_obj)] -> list of bools."""
        short = [len(ps) != self.n_public for ps, _ in items]
        pr = b"".join(plonk_proof_from_obj(p) for _, p in items)
        pub = b"".join(b"".join(_le32(x) for x in (ps if not s else [0] * self.n_public)) for (ps, _), s in zip(items, short))
        ok = C.create_string_buffer(max(1, len(items)))
        _check(load().g16_plonk_verify_batch(self._h, pr, pub, len(items), ok))
        return [ok.raw[i] != 0 and not short[i] for i in range(len(items))]

    def verify(self, public_signals, proof):
        return self.verify_batch([(public_signals, proof)])[0]

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().g16_plonk_verifier_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def plonk_setup(r1cs, seed, device=0, with_lagrange=True):
    """Test-only PLONK setup with a known tau: .r1cs bytes -> snarkjs-layout PLONK .zkey bytes."""
    z, zl = C.c_void_p(), C.c_size_t()
    _check(load().g16_plonk_setup(r1cs, len(r1cs), seed, device, 1 if with_lagrange else 0, C.byref(z), C.byref(zl)))
    return _take(z, zl)


def plonk_setup_ptau(r1cs, ptau, device=0, with_lagrange=True):
    """`snarkjs plonk setup c.r1cs pot.ptau c.zkey`: .r1cs and .ptau bytes -> PLONK .zkey bytes."""
    z, zl = C.c_void_p(), C.c_size_t()
    _check(load().g16_plonk_setup_ptau(r1cs, len(r1cs), ptau, len(ptau), device, 1 if with_lagrange else 0, C.byref(z), C.byref(zl)))
    return _take(z, zl)


def plonk_proof_to_obj(pr):
    """g16_plonk_proof -> the object `snarkjs plonk prove` stringifies (its key order)."""
    def g1(b):
        b = bytes(b)
        return ["0", "1", "0"] if b == bytes(64) else [_dec(b[:32]), _dec(b[32:]), "1"]
    o = {}
    for k in _PLONK_G1:
        o[k] = g1(getattr(pr, k))
    for k in _PLONK_EV:
        o[k] = _dec(bytes(getattr(pr, k)))
    o["Wxi"], o["Wxiw"] = g1(pr.Wxi), g1(pr.Wxiw)
    o["protocol"], o["curve"] = "plonk", "bn128"
    return o


def pairing_op(pairs, device=0):
    """[(G1 affine ints, G2 affine ((x0, x1), (y0, y1)))] -> per pair the 12 tower coordinates (ints) of the pairing
    value the verifier kernels compute."""
    buf = b"".join(_le32(P[0]) + _le32(P[1]) + _le32(Q[0][0]) + _le32(Q[0][1]) + _le32(Q[1][0]) + _le32(Q[1][1]) for P, Q in pairs)
    out = C.create_string_buffer(max(1, 384 * len(pairs)))
    _check(load().g16_pairing_op(device, buf, len(pairs), out))
    raw = out.raw
    return [[int.from_bytes(raw[384 * i + 32 * k:384 * i + 32 * k + 32], "little") for k in range(12)] for i in range(len(pairs))]


def synth_witness(n_vars, n_public, n_constraints, seed, wseed):
    lib = load()
    w, wl = C.c_void_p(), C.c_size_t()
    _check(lib.g16_synth_witness(n_vars, n_public, n_constraints, seed, wseed, C.byref(w), C.byref(wl)))
    return _take(w, wl)
