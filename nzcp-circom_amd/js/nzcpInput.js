// NZ COVID Pass -> circuit input / expected public signals (SURVEY.md 8f row 1).
//
// Restates what the reference's test harness does around the circuit, so that a prover run can be
// fed from a pass URI and its public.json cross-checked against the pass itself:
//   - /root/reference/test/helpers/nzcp.js:140-158  pass URI -> COSE_Sign1 -> Sig_structure
//     ("ToBeSigned": ["Signature1", protected, h'', payload] as CBOR)
//   - /root/reference/test/nzcp.js:11-16,36-38      zero-pad to MaxToBeSignedBytes, bits MSB-first
//     per byte (test/helpers/utils.js:2-10), toBeSignedLen = byte length
//   - /root/reference/test/nzcp.js:18-30,41-48      expected public signals: sha256("given,family,dob")
//     bits | sha256(ToBeSigned) bits | exp   (nPublic = 513; circuit outputs nzcptpl.circom:466-468)
// Written from the pass specification (base32 RFC 4648, CBOR RFC 8949, COSE_Sign1 RFC 8152, CWT
// claim 4 = exp); no npm dependencies.
"use strict";
const crypto = require("crypto");

const B32 = "ABCDEFGHIJKLMNOPQRSTUVWXYZ234567";

function base32Decode(text) {
  const out = [];
  let acc = 0, nbits = 0;
  for (const ch of text.replace(/=+$/, "")) {
    const v = B32.indexOf(ch);
    if (v < 0) throw new Error("pass URI: invalid base32 character");
    acc = ((acc << 5) | v) & 0xfff;
    nbits += 5;
    if (nbits >= 8) {
      nbits -= 8;
      out.push((acc >> nbits) & 0xff);
    }
  }
  return Buffer.from(out);
}

// Minimal CBOR reader: enough for COSE_Sign1 and the CWT claims of a pass.
class Cbor {
  constructor(buf) { this.b = buf; this.p = 0; }
  byte() { if (this.p >= this.b.length) throw new Error("CBOR: truncated"); return this.b[this.p++]; }
  arg(info) {
    if (info < 24) return info;
    const n = { 24: 1, 25: 2, 26: 4, 27: 8 }[info];
    if (!n) throw new Error("CBOR: unsupported length encoding");
    let v = 0;
    for (let i = 0; i < n; i++) v = v * 256 + this.byte();
    return v;
  }
  item() {
    const head = this.byte(), major = head >> 5, a = this.arg(head & 31);
    switch (major) {
      case 0: return a;
      case 1: return -1 - a;
      case 2: { const s = this.b.slice(this.p, this.p + a); if (s.length !== a) throw new Error("CBOR: truncated"); this.p += a; return s; }
      case 3: { const s = this.b.slice(this.p, this.p + a); if (s.length !== a) throw new Error("CBOR: truncated"); this.p += a; return s.toString("utf-8"); }
      case 4: { const arr = []; for (let i = 0; i < a; i++) arr.push(this.item()); return arr; }
      case 5: { const m = new Map(); for (let i = 0; i < a; i++) { const k = this.item(); m.set(k, this.item()); } return m; }
      case 6: return { tag: a, value: this.item() };
      default: throw new Error("CBOR: unsupported major type 7");
    }
  }
}

function bstrHeader(len) {
  if (len < 24) return Buffer.from([0x40 + len]);
  if (len < 256) return Buffer.from([0x58, len]);
  if (len < 65536) return Buffer.from([0x59, len >> 8, len & 0xff]);
  throw new Error("byte string too long");
}

function parsePassURI(uri) {
  const m = /^NZCP:\/(\d+)\/([A-Z2-7=]+)$/.exec(uri.trim());
  if (!m) throw new Error("not an NZCP pass URI");
  if (m[1] !== "1") throw new Error("unsupported NZCP version " + m[1]);
  const top = new Cbor(base32Decode(m[2])).item();
  if (!top || top.tag !== 18 || !Array.isArray(top.value) || top.value.length !== 4) throw new Error("not a COSE_Sign1 structure");
  const [protectedHdr, , payload, signature] = top.value;
  if (!Buffer.isBuffer(protectedHdr) || !Buffer.isBuffer(payload) || !Buffer.isBuffer(signature)) throw new Error("malformed COSE_Sign1");
  return { protectedHdr, payload, signature };
}

// Sig_structure = [ "Signature1", body_protected, external_aad = h'', payload ]
function toBeSigned(uri) {
  const { protectedHdr, payload } = parsePassURI(uri);
  const ctx = Buffer.from("Signature1", "ascii");
  return Buffer.concat([Buffer.from([0x84, 0x60 + ctx.length]), ctx,
                        bstrHeader(protectedHdr.length), protectedHdr, bstrHeader(0),
                        bstrHeader(payload.length), payload]);
}

function bitsMsbFirst(buf) {
  const bits = new Array(buf.length * 8);
  for (let i = 0; i < buf.length; i++) for (let k = 0; k < 8; k++) bits[i * 8 + k] = (buf[i] >> (7 - k)) & 1;
  return bits;
}

// circuit input of NZCPPubIdentity (nzcptpl.circom:464-465): maxBytes = 314 (example) / 355 (live)
function circuitInput(uri, maxBytes) {
  const tbs = toBeSigned(uri);
  if (tbs.length > maxBytes) throw new Error(`ToBeSigned is ${tbs.length} bytes, circuit maximum is ${maxBytes}`);
  const padded = Buffer.alloc(maxBytes);
  tbs.copy(padded);
  return { toBeSigned: bitsMsbFirst(padded), toBeSignedLen: tbs.length };
}

function claims(uri) {
  const cwt = new Cbor(parsePassURI(uri).payload).item();
  if (!(cwt instanceof Map)) throw new Error("CWT claims are not a map");
  const vc = cwt.get("vc");
  const subj = vc instanceof Map ? vc.get("credentialSubject") : undefined;
  if (!(subj instanceof Map)) throw new Error("no credentialSubject in the pass");
  return { exp: cwt.get(4), nbf: cwt.get(5), iss: cwt.get(1),
           givenName: subj.get("givenName"), familyName: subj.get("familyName"), dob: subj.get("dob") };
}

// what public.json must contain for this pass: 256 + 256 bits (MSB-first) then exp, decimal strings
function expectedPublicSignals(uri) {
  const c = claims(uri);
  const credSubj = `${c.givenName},${c.familyName},${c.dob}`;
  const h1 = crypto.createHash("sha256").update(credSubj, "utf-8").digest();
  const h2 = crypto.createHash("sha256").update(toBeSigned(uri)).digest();
  return bitsMsbFirst(h1).concat(bitsMsbFirst(h2)).map(String).concat([String(c.exp)]);
}

// convenience for `groth16.prove` users: does a public.json match the pass it claims to be about?
function publicSignalsMatchPass(publicSignals, uri) {
  const want = expectedPublicSignals(uri);
  return publicSignals.length === want.length && publicSignals.every((v, i) => String(v) === want[i]);
}

module.exports = { base32Decode, parsePassURI, toBeSigned, circuitInput, claims, expectedPublicSignals, publicSignalsMatchPass };
