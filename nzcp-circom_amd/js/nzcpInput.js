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

// ---------------------------------------------------------------- synthetic passes (SURVEY 8f row 1, second half)
// Real live passes are private (/root/reference/.env.example), so the live FORMAT is generated: same CWT
// structure as the MoH example pass, with the live issuer "did:web:nzcp.identity.health.nz" (31 chars) and an
// 8-byte kid -- which is what moves the claims map to offset 30 of ToBeSigned (ClaimsSkip,
// /root/reference/circuits/nzcptpl.circom:438) and exp / vc / credentialSubject to 72 / 80 / 250
// (/root/reference/test/nzcp.js:103,158,230).  The signature is random bytes: the circuit never checks it
// (/root/reference/README.md:19-22).
function base32Encode(buf) {
  let out = "", acc = 0, nbits = 0;
  for (const byte of buf) {
    acc = (acc << 8) | byte;
    nbits += 8;
    while (nbits >= 5) { nbits -= 5; out += B32[(acc >> nbits) & 31]; }
    acc &= (1 << nbits) - 1;
  }
  if (nbits > 0) out += B32[(acc << (5 - nbits)) & 31];
  return out;
}
function cborHead(major, n) {
  if (n < 24) return Buffer.from([(major << 5) | n]);
  if (n < 256) return Buffer.from([(major << 5) | 24, n]);
  if (n < 65536) return Buffer.from([(major << 5) | 25, n >> 8, n & 0xff]);
  return Buffer.from([(major << 5) | 26, (n >>> 24) & 0xff, (n >>> 16) & 0xff, (n >>> 8) & 0xff, n & 0xff]);
}
const cborText = (t) => { const b = Buffer.from(t, "utf-8"); return Buffer.concat([cborHead(3, b.length), b]); };
const cborBytes = (b) => Buffer.concat([cborHead(2, b.length), b]);
const cborU32 = (v) => Buffer.from([0x1a, (v >>> 24) & 0xff, (v >>> 16) & 0xff, (v >>> 8) & 0xff, v & 0xff]);

function syntheticPass(opts) {
  const o = Object.assign({ format: "live", givenName: "Jack", familyName: "Sparrow", dob: "1960-04-16",
                            nbf: 1635883530, exp: 1951416330 }, opts || {});
  const live = o.format === "live";
  const iss = o.iss || (live ? "did:web:nzcp.identity.health.nz" : "did:web:nzcp.covid19.health.nz");
  const kid = o.kid || (live ? Buffer.from("z12Kf7UQ", "ascii") : Buffer.from("key-1", "ascii"));
  const cti = o.cti || crypto.createHash("sha256").update(`${o.givenName}|${o.familyName}|${o.dob}`).digest().slice(0, 16);
  const signature = o.signature || crypto.createHash("sha512").update(cti).digest();
  const protectedHdr = Buffer.concat([Buffer.from([0xa2, 0x04]), cborBytes(kid), Buffer.from([0x01, 0x26])]);
  const subj = Buffer.concat([Buffer.from([0xa3]), cborText("givenName"), cborText(o.givenName), cborText("familyName"),
                              cborText(o.familyName), cborText("dob"), cborText(o.dob)]);
  const vc = Buffer.concat([Buffer.from([0xa4]), cborText("@context"), Buffer.from([0x82]),
                            cborText("https://www.w3.org/2018/credentials/v1"), cborText("https://nzcp.covid19.health.nz/contexts/v1"),
                            cborText("version"), cborText("1.0.0"), cborText("type"), Buffer.from([0x82]),
                            cborText("VerifiableCredential"), cborText("PublicCovidPass"), cborText("credentialSubject"), subj]);
  const payload = Buffer.concat([Buffer.from([0xa5, 0x01]), cborText(iss), Buffer.from([0x05]), cborU32(o.nbf), Buffer.from([0x04]),
                                 cborU32(o.exp), cborText("vc"), vc, Buffer.from([0x07]), cborBytes(cti)]);
  const cose = Buffer.concat([Buffer.from([0xd2, 0x84]), cborBytes(protectedHdr), Buffer.from([0xa0]), cborBytes(payload),
                              cborBytes(signature)]);
  return "NZCP:/1/" + base32Encode(cose);
}

// Byte offsets inside ToBeSigned of the three credential strings and of the 4 exp bytes: the circuit constants
// of the fixed-layout NZCP interface circuit (g16_nzcp_fixed_layout_setup); also the offsets the reference's
// tests name (claims map, exp, vc, credentialSubject).
function fixedLayout(uri) {
  const tbs = toBeSigned(uri);
  const find = (needle, from) => { const i = tbs.indexOf(needle, from || 0); if (i < 0) throw new Error("layout: key not found"); return i; };
  const textAt = (pos) => {   // CBOR text string header at pos -> [offset of the characters, length]
    const r = new Cbor(tbs); r.p = pos;
    const head = r.byte();
    if (head >> 5 !== 3) throw new Error("layout: expected a text string");
    const len = r.arg(head & 31);
    return [r.p, len];
  };
  const { payload } = parsePassURI(uri);
  const claimsAt = tbs.length - payload.length;
  const subjKey = find(cborText("credentialSubject"));
  const given = textAt(find(cborText("givenName"), subjKey) + 10);
  const family = textAt(find(cborText("familyName"), subjKey) + 11);
  const dob = textAt(find(cborText("dob"), subjKey) + 4);
  // claim 4 (exp): walk the top-level claims map
  const r = new Cbor(tbs); r.p = claimsAt;
  const head = r.byte();
  if (head >> 5 !== 5) throw new Error("layout: claims are not a map");
  let expOff = -1;
  for (let i = 0, n = r.arg(head & 31); i < n; i++) {
    const k = r.item();
    if (k === 4) { if (tbs[r.p] !== 0x1a) throw new Error("layout: exp is not a 32-bit uint"); expOff = r.p + 1; }
    r.item();
  }
  if (expOff < 0) throw new Error("layout: no exp claim");
  return { toBeSignedLen: tbs.length, claimsAt, expAt: expOff - 1, expOff, vcAt: find(cborText("vc")) + 3,
           credSubjAt: subjKey + 18, segs: [given, family, dob] };
}

module.exports = { base32Decode, base32Encode, parsePassURI, toBeSigned, circuitInput, claims, expectedPublicSignals,
                   publicSignalsMatchPass, syntheticPass, fixedLayout };
