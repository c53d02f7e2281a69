// snarkjs-shaped host API over the MI355X prover (Node.js >= 12, N-API addon, no npm deps).
//
// Drop-in for the prove path of snarkjs 0.4.12 (pin /root/reference/yarn.lock:987-1001;
// /root/reference/package.json:12):
//     const { groth16 } = require("nzcp-circom_amd/js");
//     const { proof, publicSignals } = await groth16.prove(zkey, wtns);
// zkey / wtns: a path, a Buffer/Uint8Array, or snarkjs's {type: "mem", data: Uint8Array}.
// proof / publicSignals have exactly snarkjs's shape (decimal strings; key order pi_a, pi_b, pi_c,
// protocol, curve), so JSON.stringify(x, null, 1) reproduces proof.json / public.json byte for byte
// given the same blinding (r, s).  Errors are thrown Errors with snarkjs's messages.
// groth16.verify(vk, publicSignals, proof) is snarkjs's too (GPU pairing check, csrc/verify.hip); createVerifier(vk)
// keeps the key resident and verifies batches, one verdict per proof.
// Extensions (not in snarkjs): opts = {r, s, device, devices, windowBits}; groth16.createProver() keeps the
// proving key resident in HBM across proofs; prover.proveBatch(wtnsList); createProver(zkey, {devices: [0, 1, ...]})
// shards ONE proof over several GPUs of the node (BASELINE config 4: MSM point ranges + the A/B/C evaluation split
// across the devices, partial sums added on the host).
"use strict";
const fs = require("fs");
const path = require("path");

let addon = null;
function native() {
  if (!addon) {
    const p = path.join(__dirname, "addon", "g16_napi.node");
    if (!fs.existsSync(p)) throw new Error(`${p} not built (run make -C ${path.dirname(p)}); there is no JS fallback`);
    addon = require(p);
  }
  return addon;
}

function toBuffer(x, what) {
  if (typeof x === "string") return fs.readFileSync(x);
  if (Buffer.isBuffer(x)) return x;
  if (x instanceof Uint8Array) return Buffer.from(x.buffer, x.byteOffset, x.byteLength);
  if (x && x.type === "mem" && x.data) return toBuffer(x.data, what);
  throw new Error(`${what}: expected a path, a Buffer/Uint8Array or {type:"mem", data}`);
}

function scalarToBuffer(v) {
  if (v === undefined || v === null) return null;
  if (Buffer.isBuffer(v)) return v;
  let n = BigInt(v);
  const out = Buffer.alloc(32);
  for (let i = 0; i < 32; i++) { out[i] = Number(n & 0xffn); n >>= 8n; }
  return out;
}

function dec(buf, off) {
  let n = 0n;
  for (let i = 31; i >= 0; i--) n = (n << 8n) | BigInt(buf[off + i]);
  return n.toString();
}
function isZero(buf, off, len) {
  for (let i = 0; i < len; i++) if (buf[off + i]) return false;
  return true;
}

// g16_proof bytes -> what snarkjs's G1/G2.toObject + stringifyBigInts produce
function proofObject(raw) {
  const g1 = (o) => (isZero(raw, o, 64) ? ["0", "1", "0"] : [dec(raw, o), dec(raw, o + 32), "1"]);
  const pi_b = isZero(raw, 64, 128)
    ? [["0", "0"], ["1", "0"], ["0", "0"]]
    : [[dec(raw, 64), dec(raw, 96)], [dec(raw, 128), dec(raw, 160)], ["1", "0"]];
  return { pi_a: g1(0), pi_b, pi_c: g1(192), protocol: "groth16", curve: "bn128" };
}
function publicSignals(pub) {
  const out = [];
  for (let o = 0; o < pub.length; o += 32) out.push(dec(pub, o));
  return out;
}

class Prover {
  constructor(handle) { this._h = handle; this._busy = Promise.resolve(); }
  get info() { return native().info(this._h); }
  get timings() { return native().timings(this._h); }
  // one in-flight prove per handle: serialise callers
  prove(wtns, opts = {}) {
    if (!this._h) return Promise.reject(new Error("prover is closed"));
    const w = toBuffer(wtns, "wtns"), h = this._h;
    const run = () => native().prove(h, w, scalarToBuffer(opts.r), scalarToBuffer(opts.s))
      .then(({ proof, pub }) => ({ proof: proofObject(proof), publicSignals: publicSignals(pub) }));
    const p = this._busy.then(run, run);
    this._busy = p.catch(() => {});
    return p;
  }
  // Batch of independent witnesses against the resident key (g16_prove_batch: the library overlaps
  // proof i+1's device work with proof i's tail).  opts.r / opts.s are TEST-ONLY: they pin ONE blinding pair
  // for every proof of the batch (reproducible bytes), which breaks zero-knowledge across the batch -- leave
  // them unset in production and every proof draws its own (r, s) from the OS CSPRNG.
  proveBatch(wtnsList, opts = {}) {
    if (!this._h) return Promise.reject(new Error("prover is closed"));
    const ws = wtnsList.map((w) => toBuffer(w, "wtns"));
    let rs = null;
    const r = scalarToBuffer(opts.r), s = scalarToBuffer(opts.s);
    if (r && s) {
      rs = Buffer.alloc(ws.length * 64);
      for (let i = 0; i < ws.length; i++) { r.copy(rs, i * 64); s.copy(rs, i * 64 + 32); }
    }
    const h = this._h;
    const run = () => native().proveBatch(h, ws, rs)
      .then((list) => list.map(({ proof, pub }) => ({ proof: proofObject(proof), publicSignals: publicSignals(pub) })));
    const p = this._busy.then(run, run);
    this._busy = p.catch(() => {});
    return p;
  }
  // Releases the resident key.  Proofs already requested (awaited or not) finish first: the native handle is
  // destroyed after the last of them settles, never under a running proof.  Returns a Promise.
  close() {
    if (!this._h) return this._busy;
    const h = this._h;
    this._h = null;
    this._busy = this._busy.then(() => native().destroy(h));
    return this._busy;
  }
}

async function createProver(zkey, opts = {}) {
  const z = toBuffer(zkey, "zkey");
  if (opts.shardCount > 1 || opts.shardRank)
    throw new Error("createProver: shardRank/shardCount belong to the C ABI of multi-process hosts; in Node pass " +
                    "devices: [ordinal, ...] and the proof is sharded over those GPUs inside this process");
  const nopts = { device: opts.device | 0, windowBits: opts.windowBits | 0, taskLen: opts.taskLen | 0 };
  if (Array.isArray(opts.devices) && opts.devices.length) nopts.devices = opts.devices.map((d) => d | 0);
  const h = await native().create(z, nopts);
  return new Prover(h);
}

// ------------------------------------------------------------------ verifier (snarkjs groth16.verify, batched on the GPU)
function le32(v) {
  let n = BigInt(v);
  if (n < 0n) throw new Error("negative field element");
  const out = Buffer.alloc(32);
  for (let i = 0; i < 32; i++) { out[i] = Number(n & 0xffn); n >>= 8n; }
  return out;   // (values >= 2^256 are cut: no field element is that large)
}
function g1Bytes(t) { return String(t[2]) === "0" ? Buffer.alloc(64) : Buffer.concat([le32(t[0]), le32(t[1])]); }
function g2Bytes(t) {
  if (String(t[2][0]) === "0" && String(t[2][1]) === "0") return Buffer.alloc(128);
  return Buffer.concat([le32(t[0][0]), le32(t[0][1]), le32(t[1][0]), le32(t[1][1])]);
}
// proof.json object -> g16_proof bytes; verification_key.json object -> the C ABI's point image (standard form)
function proofBytes(p) { return Buffer.concat([g1Bytes(p.pi_a), g2Bytes(p.pi_b), g1Bytes(p.pi_c)]); }
function vkeyBytes(vk) {
  if (!vk || vk.protocol !== "groth16" || !Array.isArray(vk.IC) || vk.IC.length !== Number(vk.nPublic) + 1)
    throw new Error("verification key: not a groth16 key (protocol / IC / nPublic)");
  return Buffer.concat([g1Bytes(vk.vk_alpha_1), g2Bytes(vk.vk_beta_2), g2Bytes(vk.vk_gamma_2), g2Bytes(vk.vk_delta_2),
    ...vk.IC.map(g1Bytes)]);
}

class Verifier {
  constructor(handle, nPublic) { this._h = handle; this.nPublic = nPublic; }
  // items: [{publicSignals, proof}] -> Promise<boolean[]>, one verdict per proof (g16_verify_batch)
  async verifyBatch(items) {
    if (!this._h) throw new Error("verifier is closed");
    const short = items.map((it) => it.publicSignals.length !== this.nPublic);
    const proofs = Buffer.concat(items.map((it) => proofBytes(it.proof)));
    const pubs = Buffer.concat(items.map((it, i) => (short[i] ? Buffer.alloc(this.nPublic * 32)
      : Buffer.concat(it.publicSignals.map(le32)))));
    const ok = await native().verifyBatch(this._h, proofs, pubs);
    return items.map((_, i) => !short[i] && ok[i] !== 0);
  }
  async verify(publicSignals, proof) { return (await this.verifyBatch([{ publicSignals, proof }]))[0]; }
  close() { if (this._h) { native().destroyVerifier(this._h); this._h = null; } }
}

async function createVerifier(vk, opts = {}) {
  const h = await native().createVerifier(vkeyBytes(vk), Number(vk.nPublic), 0, opts.device | 0);
  return new Verifier(h, Number(vk.nPublic));
}

// ------------------------------------------------------------------ PLONK (snarkjs plonk.prove on the GPU, csrc/plonk.hip)
const PLONK_G1 = ["A", "B", "C", "Z", "T1", "T2", "T3"];
const PLONK_EV = ["eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw", "eval_r"];
// g16_plonk_proof bytes -> the object snarkjs's plonk_prove.js stringifies (its key order)
function plonkProofObject(raw) {
  const g1 = (o) => (isZero(raw, o, 64) ? ["0", "1", "0"] : [dec(raw, o), dec(raw, o + 32), "1"]);
  const out = {};
  let o = 0;
  for (const k of PLONK_G1) { out[k] = g1(o); o += 64; }
  for (const k of PLONK_EV) { out[k] = dec(raw, o); o += 32; }
  out.Wxi = g1(o); out.Wxiw = g1(o + 64);
  out.protocol = "plonk"; out.curve = "bn128";
  return out;
}
class PlonkProver {
  constructor(handle) { this._h = handle; this._busy = Promise.resolve(); }
  // opts.blinding: nine scalars b1..b9 (decimal strings / BigInts) for a reproducible proof; default: fresh randomness
  prove(wtns, opts = {}) {
    if (!this._h) return Promise.reject(new Error("prover is closed"));
    const w = toBuffer(wtns, "wtns"), h = this._h;
    let bl = null;
    if (opts.blinding) {
      if (opts.blinding.length !== 9) return Promise.reject(new Error("blinding: nine scalars b1..b9 expected"));
      bl = Buffer.concat(opts.blinding.map(scalarToBuffer));
    }
    const run = () => native().plonkProve(h, w, bl)
      .then(({ proof, pub }) => ({ proof: plonkProofObject(proof), publicSignals: publicSignals(pub) }));
    const p = this._busy.then(run, run);
    this._busy = p.catch(() => {});
    return p;
  }
  close() {
    if (!this._h) return this._busy;
    const h = this._h;
    this._h = null;
    this._busy = this._busy.then(() => native().plonkDestroy(h));
    return this._busy;
  }
}
// verification_key.json / proof.json of PLONK -> the C ABI images (g16_plonk_verifier_create, g16_plonk_proof)
function plonkVkeyBytes(vk) {
  if (!vk || vk.protocol !== "plonk") throw new Error("verification key: not a plonk key");
  const head = Buffer.alloc(8);
  head.writeUInt32LE(Number(vk.power), 0);
  head.writeUInt32LE(Number(vk.nPublic), 4);
  return Buffer.concat([head, le32(vk.k1), le32(vk.k2), ...["Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"].map((k) => g1Bytes(vk[k])),
    g2Bytes(vk.X_2)]);
}
function plonkProofBytes(p) {
  return Buffer.concat([...PLONK_G1.map((k) => g1Bytes(p[k])), ...PLONK_EV.map((k) => le32(p[k])), g1Bytes(p.Wxi), g1Bytes(p.Wxiw)]);
}
class PlonkVerifier {
  constructor(handle, nPublic) { this._h = handle; this.nPublic = nPublic; }
  async verifyBatch(items) {
    if (!this._h) throw new Error("verifier is closed");
    const short = items.map((it) => it.publicSignals.length !== this.nPublic);
    const proofs = Buffer.concat(items.map((it) => plonkProofBytes(it.proof)));
    const pubs = Buffer.concat(items.map((it, i) => (short[i] ? Buffer.alloc(this.nPublic * 32)
      : Buffer.concat(it.publicSignals.map(le32)))));
    const ok = await native().verifyBatch(this._h, proofs, pubs);
    return items.map((_, i) => !short[i] && ok[i] !== 0);
  }
  async verify(publicSignals, proof) { return (await this.verifyBatch([{ publicSignals, proof }]))[0]; }
  close() { if (this._h) { native().destroyVerifier(this._h); this._h = null; } }
}

const plonk = {
  // snarkjs: plonk.verify(vk_verifier, publicSignals, proof[, logger]) -> boolean
  async verify(vk, publicSignals, proof, opts = {}) {
    const v = await plonk.createVerifier(vk, opts && typeof opts.debug === "function" ? {} : opts);
    try {
      return await v.verify(publicSignals, proof);
    } finally {
      v.close();
    }
  },
  async createVerifier(vk, opts = {}) {
    const h = await native().createVerifier(plonkVkeyBytes(vk), Number(vk.nPublic), 0, opts.device | 0, 1);
    return new PlonkVerifier(h, Number(vk.nPublic));
  },
  // snarkjs: plonk.prove(zkeyFileName, witnessFileName[, logger]) -> {proof, publicSignals}
  async prove(zkey, wtns, opts = {}) {
    if (opts && typeof opts.debug === "function") opts = {};
    const prover = await plonk.createProver(zkey, opts);
    try {
      return await prover.prove(wtns, opts);
    } finally {
      await prover.close();
    }
  },
  async createProver(zkey, opts = {}) {
    return new PlonkProver(await native().plonkCreate(toBuffer(zkey, "zkey"), opts.device | 0));
  },
  // snarkjs: plonk.setup(r1csName, ptauName, zkeyName[, logger]) -- file names (a ceremony file exceeds a Buffer).
  // opts.lagrange: also write the Lagrange section 13 (snarkjs's own prover reads it; this one does not -- it is
  // nPublic x 5N field elements, 344 GB for 513 public signals at N = 2^22).
  async setup(r1csName, ptauName, zkeyName, opts = {}) {
    if (opts && typeof opts.debug === "function") opts = {};
    await native().plonkSetupFiles(String(r1csName), String(ptauName), String(zkeyName), opts.device | 0, opts.lagrange ? 1 : 0);
  },
};

// ------------------------------------------------------------------ zkey export verificationkey (host-only: header reads)
// snarkjs `zKey.exportVerificationKey(zkey)` / CLI `zkey export verificationkey <zkey> <vk.json>` -- the second line of
// the reference's PLONK flow (/root/reference/Makefile:32).  Groth16 and PLONK keys; points leave Montgomery form here
// (BigInt arithmetic, a few hundred values).  Groth16: `vk_alphabeta_12` -- redundant, snarkjs's verifier does not
// read it -- is not written.
const FQ = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
function modpow(b, e, m) { let r = 1n; b %= m; while (e > 0n) { if (e & 1n) r = (r * b) % m; b = (b * b) % m; e >>= 1n; } return r; }
const RQ_INV = modpow((1n << 256n) % FQ, FQ - 2n, FQ), RR_INV = modpow((1n << 256n) % FR, FR - 2n, FR);
function leBig(buf, off) { let n = 0n; for (let i = 31; i >= 0; i--) n = (n << 8n) | BigInt(buf[off + i]); return n; }
function sections(buf, what) {
  if (buf.length < 12 || buf.toString("latin1", 0, 4) !== what) throw new Error(`${what}: Invalid File format`);
  const n = buf.readUInt32LE(8), out = {};
  let pos = 12;
  for (let i = 0; i < n; i++) {
    if (pos + 12 > buf.length) throw new Error(`${what}: Invalid File format`);
    const id = buf.readUInt32LE(pos), size = Number(buf.readBigUInt64LE(pos + 4));
    pos += 12;
    if (size > buf.length - pos) throw new Error(`${what}: Invalid File format`);
    if (!(id in out)) out[id] = { pos, size };
    pos += size;
  }
  return out;
}
function exportVerificationKey(zkey) {
  const b = toBuffer(zkey, "zkey"), s = sections(b, "zkey");
  if (!s[1] || !s[2]) throw new Error("zkey: Invalid File format");
  const proto = b.readUInt32LE(s[1].pos);
  const fq = (o) => ((leBig(b, o) * RQ_INV) % FQ).toString();
  const g1 = (o) => (isZero(b, o, 64) ? ["0", "1", "0"] : [fq(o), fq(o + 32), "1"]);
  const g2 = (o) => (isZero(b, o, 128) ? [["0", "0"], ["1", "0"], ["0", "0"]]
    : [[fq(o), fq(o + 32)], [fq(o + 64), fq(o + 96)], ["1", "0"]]);
  let h = s[2].pos + 72;   // past n8q, q, n8r, r
  if (proto === 1) {
    const nPublic = b.readUInt32LE(h + 4);
    h += 12;
    const vk = { protocol: "groth16", curve: "bn128", nPublic, vk_alpha_1: g1(h), vk_beta_2: g2(h + 128), vk_gamma_2: g2(h + 256),
      vk_delta_2: g2(h + 448), IC: [] };
    if (!s[3] || s[3].size !== (nPublic + 1) * 64) throw new Error("zkey: Invalid File format");
    for (let i = 0; i <= nPublic; i++) vk.IC.push(g1(s[3].pos + 64 * i));
    return vk;
  }
  if (proto === 2) {
    const nPublic = b.readUInt32LE(h + 4), domainSize = b.readUInt32LE(h + 8);
    const power = 31 - Math.clz32(domainSize);
    h += 20;
    const fr = (o) => ((leBig(b, o) * RR_INV) % FR).toString();
    const vk = { protocol: "plonk", curve: "bn128", nPublic, power, k1: fr(h), k2: fr(h + 32) };
    h += 64;
    for (const k of ["Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"]) { vk[k] = g1(h); h += 64; }
    vk.X_2 = g2(h);
    vk.w = modpow(5n, (FR - 1n) >> BigInt(power), FR).toString();   // Fr.w[power]: 5^((r-1)/2^power)
    return vk;
  }
  throw new Error("zkey: unknown protocol");
}

const groth16 = {
  // snarkjs: groth16.verify(vk_verifier, publicSignals, proof[, logger]) -> boolean
  async verify(vk, publicSignals, proof, opts = {}) {
    const v = await createVerifier(vk, opts && typeof opts.debug === "function" ? {} : opts);
    try {
      return await v.verify(publicSignals, proof);
    } finally {
      v.close();
    }
  },
  createVerifier,
  // snarkjs: groth16.prove(zkeyFileName, witnessFileName[, logger]) -> {proof, publicSignals}
  async prove(zkey, wtns, opts = {}) {
    if (opts && typeof opts.debug === "function") opts = { logger: opts };   // snarkjs passes a logger third
    const prover = await createProver(zkey, opts);
    try {
      return await prover.prove(wtns, opts);
    } finally {
      await prover.close();
    }
  },
  createProver,
};

module.exports = { groth16, plonk, zKey: { exportVerificationKey }, exportVerificationKey, PlonkProver, plonkProofObject, createProver, Prover, createVerifier, Verifier, proofObject, publicSignals, proofBytes, vkeyBytes };
