#!/usr/bin/env node
// CLI twin of `snarkjs groth16 prove <circuit.zkey> <witness.wtns> <proof.json> <public.json>`
// (snarkjs cli.js groth16Prove [EXT]; the reference's Makefile scripts the sibling PLONK lines,
// /root/reference/Makefile:30-33).  Writes JSON.stringify(x, null, 1) like snarkjs.
// And of `snarkjs groth16 verify <verification_key.json> <public.json> <proof.json>` (groth16Verify [EXT]): prints
// "[INFO]  snarkJS: OK!" and exits 0, or "[ERROR] snarkJS: Invalid proof" and exits 1.
"use strict";
const fs = require("fs");
const { groth16 } = require("./index.js");

async function main(argv) {
  let a = argv.slice(2);
  if (a[0] === "zkey" && a[1] === "export" && a[2] === "verificationkey") {   // snarkjs zkey export verificationkey <zkey> [vk.json]
    const { exportVerificationKey } = require("./index.js");
    const [zk, out = "verification_key.json"] = a.slice(3);
    fs.writeFileSync(out, JSON.stringify(exportVerificationKey(zk), null, 1), "utf-8");
    return;
  }
  if (a[0] === "plonk" && a[1] === "setup") {   // snarkjs plonk setup <circuit.r1cs> <powersoftau.ptau> <circuit.zkey>
    const pos = a.slice(2).filter((x) => !x.startsWith("--"));
    if (pos.length < 3) { console.error("usage: cli.js plonk setup <circuit.r1cs> <pot.ptau> <circuit.zkey> [--lagrange]"); process.exit(2); }
    const { plonk } = require("./index.js");
    await plonk.setup(pos[0], pos[1], pos[2], { lagrange: a.includes("--lagrange") });
    return;
  }
  if (a[0] === "plonk" && a[1] === "prove") {   // snarkjs plonk prove <circuit.zkey> <witness.wtns> [proof.json] [public.json]
    const [zk, wt, proofFile = "proof.json", publicFile = "public.json"] = a.slice(2).filter((x) => !x.startsWith("--"));
    const { plonk } = require("./index.js");
    const { proof, publicSignals } = await plonk.prove(zk, wt);
    fs.writeFileSync(proofFile, JSON.stringify(proof, null, 1), "utf-8");
    fs.writeFileSync(publicFile, JSON.stringify(publicSignals, null, 1), "utf-8");
    return;
  }
  if (a[0] === "groth16") a = a.slice(1);
  if (a[0] === "plonk" && a[1] === "verify") {   // snarkjs plonk verify <verification_key.json> <public.json> <proof.json>
    const { plonk } = require("./index.js");
    const [vk = "verification_key.json", pub = "public.json", proof = "proof.json"] = a.slice(2).filter((x) => !x.startsWith("--"));
    const ok = await plonk.verify(JSON.parse(fs.readFileSync(vk, "utf-8")), JSON.parse(fs.readFileSync(pub, "utf-8")),
      JSON.parse(fs.readFileSync(proof, "utf-8")));
    if (!ok) throw new Error("Invalid proof");
    console.log("[INFO]  snarkJS: OK!");
    return;
  }
  if (a[0] === "verify") {
    const pos = a.slice(1).filter((x) => !x.startsWith("--"));
    const di = a.indexOf("--device");
    const [vk = "verification_key.json", pub = "public.json", proof = "proof.json"] = pos.filter((x, i) => di < 0 || a.indexOf(x) !== di + 1);
    const ok = await groth16.verify(JSON.parse(fs.readFileSync(vk, "utf-8")), JSON.parse(fs.readFileSync(pub, "utf-8")),
      JSON.parse(fs.readFileSync(proof, "utf-8")), { device: di >= 0 ? parseInt(a[di + 1], 10) : 0 });
    if (!ok) throw new Error("Invalid proof");
    console.log("[INFO]  snarkJS: OK!");
    return;
  }
  if (a[0] === "prove") a = a.slice(1);
  const opts = {};
  const pos = [];
  for (let i = 0; i < a.length; i++) {
    if (a[i] === "--r") opts.r = a[++i];
    else if (a[i] === "--s") opts.s = a[++i];
    else if (a[i] === "--device") opts.device = parseInt(a[++i], 10);
    else pos.push(a[i]);
  }
  if (pos.length < 2) {
    console.error("usage: cli.js [groth16] prove <circuit.zkey> <witness.wtns> [proof.json] [public.json] [--r dec --s dec --device n]");
    process.exit(2);
  }
  const [zkey, wtns, proofFile = "proof.json", publicFile = "public.json"] = pos;
  const { proof, publicSignals } = await groth16.prove(zkey, wtns, opts);
  fs.writeFileSync(proofFile, JSON.stringify(proof, null, 1), "utf-8");
  fs.writeFileSync(publicFile, JSON.stringify(publicSignals, null, 1), "utf-8");
}

main(process.argv).then(() => process.exit(0), (e) => { console.error(`[ERROR] snarkJS: ${e.message}`); process.exit(1); });
