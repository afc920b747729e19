//! bbp-ref-crosscheck: runs the REFERENCE crate against this repository's golden vectors, and makes reference-made vectors
//! for this repository's verifier.  Written for this repository (it is not a copy of anything in the reference); it only calls
//! the reference's public surface: `Proof::prove` (src/blindbid/proof.rs:36-46), `Verify::new(..).verify()`
//! (src/blindbid/verify.rs:27-89), `Bid` (src/blindbid/bid.rs).
//!
//!   bbp-ref-crosscheck verify  vectors.txt            -> one line per vector: name, accept|reject|format-error
//!   bbp-ref-crosscheck make    K N out.txt             -> K reference-made proofs with an N-entry bid list, same line format
//!   bbp-ref-crosscheck layout                          -> R1CSProof::to_bytes length and first byte of a fresh proof
//!   bbp-ref-crosscheck frames  frames.txt              -> the WIRE: the reference's own TlvWriter / TlvReader against this repository's
//!                                                         frames (server/tlv.h, server/wire.h, tests/uds_client.py); see cmd_frames
//!   bbp-ref-crosscheck trace   trace.txt               -> BYTE parity of the PROVER: Proof::prove under scripted entropy (needs the one-file
//!                                                         rand patch of rand_patch/README.md) against the golden records, field by field;
//!                                                         a mismatch names the first diverging transcript step; see cmd_trace
//!
//! Line format (what tools/ref_crosscheck/export_vectors.py writes and check_reference_made.py reads), all hex, space separated:
//!   name N toggle record score z_img seed pub_0 .. pub_{N-1} [d k y y_inv]
//! `record` = R1CSProof::to_bytes() || 4 commitments || N toggle commitments (include/bbp.h).
use std::convert::{TryFrom, TryInto};
use std::env;
use std::fs;
use std::io::Write;

use bulletproofs::r1cs::R1CSProof;
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use dusk_blindbidproof::{Bid, Proof, Verify};
use dusk_tlv::{TlvReader, TlvWriter};
use sha2::{Digest, Sha512};

fn unhex(s: &str) -> Vec<u8> {
    (0..s.len() / 2).map(|i| u8::from_str_radix(&s[2 * i..2 * i + 2], 16).expect("hex")).collect()
}
fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}
fn arr32(b: &[u8]) -> [u8; 32] {
    let mut a = [0u8; 32];
    a.copy_from_slice(b);
    a
}

/// The MiMC round constants, recomputed (the reference keeps its lazy_static private: src/blindbid/mod.rs:7-24).
fn constants() -> Vec<Scalar> {
    let mut out = Vec::with_capacity(90);
    let mut hash = [0u8; 64];
    hash.copy_from_slice(Sha512::digest(b"blind bid").as_slice());
    for _ in 0..90 {
        let c = Scalar::from_bytes_mod_order_wide(&hash);
        out.push(c);
        hash.copy_from_slice(Sha512::digest(&c.to_bytes()).as_slice());
    }
    out
}

/// Native image of mimc_gadget (src/gadgets.rs:45-67): x <- (x + key + c_i)^7 for 90 rounds, then + key.
fn mimc(left: Scalar, key: Scalar, c: &[Scalar]) -> Scalar {
    let mut x = left;
    for ci in c {
        let a = x + key + ci;
        let a2 = a * a;
        let a3 = a2 * a;
        let a4 = a2 * a2;
        x = a4 * a3;
    }
    x + key
}

fn split_record(record: &[u8], n: usize) -> Option<(Vec<u8>, Vec<CompressedRistretto>, Vec<CompressedRistretto>)> {
    let tail = 32 * (4 + n);
    if record.len() < tail + 1 {
        return None;
    }
    let pl = record.len() - tail;
    let pts: Vec<CompressedRistretto> = (0..4 + n).map(|i| CompressedRistretto::from_slice(&record[pl + 32 * i..pl + 32 * i + 32])).collect();
    Some((record[..pl].to_vec(), pts[..4].to_vec(), pts[4..].to_vec()))
}

fn cmd_verify(path: &str) {
    for line in fs::read_to_string(path).expect("vectors file").lines() {
        let f: Vec<&str> = line.split_whitespace().collect();
        if f.len() < 7 {
            continue;
        }
        let (name, n) = (f[0], f[1].parse::<usize>().unwrap());
        let record = unhex(f[3]);
        let verdict = match split_record(&record, n) {
            None => "format-error",
            Some((pbytes, commitments, t_c)) => match R1CSProof::from_bytes(&pbytes) {
                Err(_) => "format-error",
                Ok(proof) => {
                    let sc = |i: usize| Scalar::from_canonical_bytes(arr32(&unhex(f[i])));
                    match (sc(4), sc(5), sc(6)) {
                        (Some(score), Some(z_img), Some(seed)) => {
                            // pub_list through Scalar::from_bits, exactly as Verify::try_from_reader_variables does (verify.rs:112-116)
                            let pubs: Vec<Scalar> = (0..n).map(|i| Scalar::from_bits(arr32(&unhex(f[7 + i])))).collect();
                            if Verify::new(proof, commitments, t_c, score, z_img, seed, pubs).verify().is_ok() { "accept" } else { "reject" }
                        }
                        _ => "format-error",
                    }
                }
            },
        };
        println!("{} {}", name, verdict);
    }
}

fn cmd_make(k: usize, n: usize, out: &str) {
    let c = constants();
    let mut rng = rand::thread_rng();
    let mut w = fs::File::create(out).expect("output file");
    for i in 0..k {
        let d = Scalar::from((rand::random::<u64>() >> 1) as u64);
        let kk = Scalar::random(&mut rng);
        let seed = Scalar::random(&mut rng);
        let m = mimc(kk, Scalar::zero(), &c);
        let x = mimc(d, m, &c);
        let y = mimc(seed, x, &c);
        let z_img = mimc(seed, m, &c);
        let y_inv = y.invert();
        let q = d * y_inv;
        let toggle = i % n;
        let mut pubs: Vec<Scalar> = (0..n).map(|_| Scalar::random(&mut rng)).collect();
        pubs[toggle] = x;
        let bids: Vec<Bid> = pubs.iter().map(|p| Bid { x: *p }).collect();
        let proof = Proof::prove(d, kk, y, y_inv, q, z_img, seed, bids, toggle as u64).expect("prove");
        let mut record = proof.proof.to_bytes();
        for p in proof.commitments.iter().chain(proof.t_c.iter()) {
            record.extend_from_slice(p.as_bytes());
        }
        let mut line = format!("ref_made_{} {} {} {} {} {} {}", i, n, toggle, hex(&record), hex(q.as_bytes()), hex(z_img.as_bytes()), hex(seed.as_bytes()));
        for p in &pubs {
            line.push(' ');
            line.push_str(&hex(p.as_bytes()));
        }
        for s in &[d, kk, y, y_inv] {
            line.push(' ');
            line.push_str(&hex(s.as_bytes()));
        }
        writeln!(w, "{}", line).unwrap();
        // the reference accepts its own proof (sanity of this harness)
        let (pbytes, commitments, t_c) = split_record(&record, n).unwrap();
        let ok = Verify::new(R1CSProof::from_bytes(&pbytes).unwrap(), commitments, t_c, q, z_img, seed, pubs.clone()).verify().is_ok();
        assert!(ok, "reference rejected its own proof");
    }
    println!("wrote {} reference-made records to {}", k, out);
}

fn cmd_layout() {
    let c = constants();
    let (d, kk, seed) = (Scalar::from(1u64), Scalar::from(2u64), Scalar::from(3u64));
    let m = mimc(kk, Scalar::zero(), &c);
    let x = mimc(d, m, &c);
    let y = mimc(seed, x, &c);
    let z_img = mimc(seed, m, &c);
    // SURVEY.md App. B witness KAT: m, x, y, z_img for d = 1, k = 2, seed = 3
    println!("m     {}\nx     {}\ny     {}\nz_img {}", hex(m.as_bytes()), hex(x.as_bytes()), hex(y.as_bytes()), hex(z_img.as_bytes()));
    println!("c[0]  {}\nc[89] {}", hex(c[0].as_bytes()), hex(c[89].as_bytes()));
    let proof = Proof::prove(d, kk, y, y.invert(), d * y.invert(), z_img, seed, vec![Bid { x }], 0).expect("prove");
    let b = proof.proof.to_bytes();
    println!("R1CSProof::to_bytes: {} bytes, first byte 0x{:02x}  (this repository assumes 1121 bytes, 0x00: SURVEY.md A.8)", b.len(), b[0]);
}

/// What the reference's own writer makes of one opcode-1 request: TlvWriter::write for each of the seven scalars (32 bytes each),
/// write_list for the bid list, write for the toggle as 8 little-endian bytes -- the element shapes src/blindbid/proof.rs:97-115
/// reads back (serde `Scalar` = 32 raw bytes, `u64`).  If the reference's Deserialize wants anything else, parsing THIS fails and
/// says so before any of this repository's bytes are looked at.
fn ref_prove_body(s7: &[[u8; 32]; 7], pubs: &[[u8; 32]], toggle: u64) -> Vec<u8> {
    let mut w = TlvWriter::new(vec![]);
    for s in s7.iter() {
        w.write(&s[..]).expect("write scalar");
    }
    let items: Vec<Vec<u8>> = pubs.iter().map(|p| p.to_vec()).collect();
    w.write_list(items.as_slice()).expect("write bid list");
    w.write(&toggle.to_le_bytes()[..]).expect("write toggle");
    w.into_inner()
}

/// ... and one opcode-2 request body (src/blindbid/verify.rs:91-128): the proof blob as ONE element, three scalars, the public list.
fn ref_verify_body(blob: &[u8], score: &[u8; 32], z_img: &[u8; 32], seed: &[u8; 32], pubs: &[[u8; 32]]) -> Vec<u8> {
    let mut w = TlvWriter::new(vec![]);
    w.write(blob).expect("write proof blob");
    for s in [score, z_img, seed].iter() {
        w.write(&s[..]).expect("write scalar");
    }
    let items: Vec<Vec<u8>> = pubs.iter().map(|p| p.to_vec()).collect();
    w.write_list(items.as_slice()).expect("write public list");
    w.into_inner()
}

/// `frames`: settles server/tlv.h (the one file of this repository that guesses dusk-tlv's bytes) in one run.  Input lines, all
/// hex, written by export_vectors.py from the golden fixtures with THIS repository's encoders:
///   wire NAME N TOGGLE  d k y y_inv q z_img seed  pub_0..pub_{N-1}  RECORD  OUR_PROVE_BODY  OUR_PROOF_BLOB  OUR_VERIFY_BODY  OUR_REPLY_FRAME
/// For every line it prints
///   1. the reference-written prove body, and whether OUR_PROVE_BODY is byte-identical to it;
///   2. whether Proof::try_from_reader_variables parses (and proves from) the reference-written body AND ours;
///   3. the reference's Proof -> Vec<u8> (TryInto, proof.rs:118-143) of the golden record, and whether OUR_PROOF_BLOB equals it;
///   4. whether Proof::try_from(OUR_PROOF_BLOB) parses and re-serialises to the same bytes;
///   5. the reference-written verify body around the reference's blob, whether OUR_VERIFY_BODY equals it, and the verdict of
///      Verify::try_from_reader_variables(..).verify() on both (must be accept);
///   6. the reply frame TlvWriter::new(socket).write(blob) produces (main.rs:89-90) against OUR_REPLY_FRAME.
/// Exit code 1 if any comparison says DIFFERENT or any parse fails.
fn cmd_frames(path: &str) {
    let mut bad = 0;
    let mut check = |what: &str, ok: bool| {
        println!("    {:<58} {}", what, if ok { "ok" } else { "DIFFERENT / FAILED" });
        if !ok {
            bad += 1;
        }
    };
    for line in fs::read_to_string(path).expect("frames file").lines() {
        let f: Vec<&str> = line.split_whitespace().collect();
        if f.len() < 12 || f[0] != "wire" {
            continue;
        }
        let (name, n, toggle) = (f[1], f[2].parse::<usize>().unwrap(), f[3].parse::<u64>().unwrap());
        let mut s7 = [[0u8; 32]; 7];
        for i in 0..7 {
            s7[i] = arr32(&unhex(f[4 + i]));
        }
        let pubs: Vec<[u8; 32]> = (0..n).map(|i| arr32(&unhex(f[11 + i]))).collect();
        let record = unhex(f[11 + n]);
        let (our_prove, our_blob, our_verify, our_reply) = (unhex(f[12 + n]), unhex(f[13 + n]), unhex(f[14 + n]), unhex(f[15 + n]));
        println!("{}", name);
        // 1, 2: prove request
        let ref_prove = ref_prove_body(&s7, &pubs, toggle);
        println!("    reference-written prove body: {}", hex(&ref_prove));
        check("our prove body == reference-written prove body", our_prove == ref_prove);
        check("reference parses + proves its own prove body", Proof::try_from_reader_variables(&ref_prove[..]).is_ok());
        check("reference parses + proves OUR prove body", Proof::try_from_reader_variables(&our_prove[..]).is_ok());
        // 3, 4: proof blob
        let (pbytes, commitments, t_c) = split_record(&record, n).expect("golden record");
        let proof = Proof::new(R1CSProof::from_bytes(&pbytes).expect("golden R1CSProof parses in the reference"), commitments, t_c);
        let ref_blob: Vec<u8> = proof.try_into().expect("Proof -> bytes");
        println!("    reference-written proof blob: {} bytes", ref_blob.len());
        check("our proof blob == reference's Proof -> Vec<u8>", our_blob == ref_blob);
        let reparsed: Result<Vec<u8>, _> = Proof::try_from(our_blob.clone()).and_then(|p| p.try_into());
        check("reference parses OUR proof blob and re-serialises it identically", reparsed.map(|b| b == our_blob).unwrap_or(false));
        // 5: verify request (score = q = s7[4], z_img = s7[5], seed = s7[6])
        let ref_verify = ref_verify_body(&ref_blob, &s7[4], &s7[5], &s7[6], &pubs);
        check("our verify body == reference-written verify body", our_verify == ref_verify);
        let verdict = |b: &[u8]| Verify::try_from_reader_variables(b).map(|v| v.verify().is_ok()).unwrap_or(false);
        check("reference accepts its own verify body", verdict(&ref_verify[..]));
        check("reference accepts OUR verify body", verdict(&our_verify[..]));
        // 6: the reply frame on the socket
        let mut w = TlvWriter::new(vec![]);
        w.write(ref_blob.as_slice()).expect("write reply");
        check("our reply frame == TlvWriter::new(socket).write(blob)", our_reply == w.into_inner());
        // and the reader side of the outer frame, as MainFuture::poll does it (main.rs:70-79)
        let mut framed = TlvWriter::new(vec![]);
        let mut req = vec![1u8];
        req.extend_from_slice(&our_prove);
        framed.write(req.as_slice()).expect("frame request");
        let framed = framed.into_inner();
        let got = TlvReader::new(&framed[..]).next().and_then(|r| r.ok());
        check("TlvReader::next() returns opcode || body from a framed request", got.map(|g| g == req).unwrap_or(false));
    }
    if bad != 0 {
        eprintln!("{} check(s) failed: correct dusk_blindbidproof_amd/server/tlv.h and tests/uds_client.py (INTEGRATION.md 2b)", bad);
        std::process::exit(1);
    }
}

/// The compact R1CSProof layout this repository assumes: version byte, 11 points, 3 scalars, then
/// the inner-product proof: 2 k points L_1 R_1 .. L_k R_k and the scalars a, b (k = 11 for the 2048-multiplier circuit).  Field names
/// are paired with the transcript step that must already agree for the field to agree: the FIRST differing field names the step.
fn proof_fields(k: usize) -> Vec<(String, usize, &'static str)> {
    let mut f: Vec<(String, usize, &'static str)> = vec![("version byte".into(), 1, "R1CSProof::to_bytes layout (phase marker)")];
    for (n, why) in [
        ("A_I1", "witness assignment a_L / a_R, generators G / H / B_blinding, blinding i~ (first draw of the TranscriptRng)"),
        ("A_O1", "a_O, blinding o~"),
        ("S1", "the TranscriptRng itself: witness rekeying order, 64-byte draws s_L / s_R in multiplier order, blinding s~"),
        ("A_I2", "identity in a one-phase proof"),
        ("A_O2", "identity in a one-phase proof"),
        ("S2", "identity in a one-phase proof"),
        ("T_1", "challenges y, z (labels, what was absorbed before them: V commitments, m, A_I1 A_O1 S1 [A_I2 A_O2 S2]), flattened constraint weights wL wR wO wV, polynomial coefficient t_1, its blinding (rng draw order)"),
        ("T_3", "t_3 and its blinding"),
        ("T_4", "t_4 and its blinding"),
        ("T_5", "t_5 and its blinding"),
        ("T_6", "t_6 and its blinding"),
    ]
    .iter()
    {
        f.push(((*n).into(), 32, *why));
    }
    f.push(("t_x".into(), 32, "challenges u (phase separator) and x, evaluation t(x)"));
    f.push(("t_x_blinding".into(), 32, "blinding polynomial at x: the wV . v_blinding term and the T_i blindings"));
    f.push(("e_blinding".into(), 32, "x (i~ + x o~ + x^2 s~) combination"));
    for j in 1..=k {
        f.push((format!("L_{}", j), 32, if j == 1 { "challenge w (Q = w B), the y^-i factors on H, l(x) / r(x) vectors, padding of the multipliers to 2048" } else { "IPA challenge u_(j-1) (label, L and R absorbed in this order) and the vector folding" }));
        f.push((format!("R_{}", j), 32, "same round: the other halves"));
    }
    f.push(("a".into(), 32, "last IPA challenge and the final fold"));
    f.push(("b".into(), 32, "last IPA challenge and the final fold"));
    f
}

/// `trace`: the PROVER, byte for byte.  Needs `rand::thread_rng()` to hand out scripted bytes (rand_patch/README.md: one function of
/// rand 0.6.5 edited, wired in with a [patch.crates-io] entry; nothing of the reference or of bulletproofs is touched).  Input lines
/// (export_vectors.py, from tests/golden/proofs_full.json):
///   trace NAME N TOGGLE  d k y y_inv q z_img seed  pub_0..pub_{N-1}  ENTROPY  RECORD  y z u x w u_1..u_k
/// ENTROPY is this repository's injected entropy: 4 + N blinding scalars (32 bytes each) then the 32-byte seed of the TranscriptRng.
/// The reference draws 64 bytes per blinding and reduces them (Scalar::random, proof.rs:53-64), then 32 bytes inside
/// Prover::prove (TranscriptRngBuilder::finalize): the script fed to the patched thread_rng is  blinding || 32 zero bytes  per
/// blinding (a canonical scalar reduces to itself), then the seed.  With that, Proof::prove is deterministic and its record must
/// equal RECORD.  On a mismatch the fields are compared in transcript order and the first differing one is reported with the step it
/// depends on and this repository's challenge values up to that step (the golden trace), so that the three implementations are
/// corrected in ONE place each: oracle/ref_py/r1cs.py, oracle/c/bbp_oracle.c, dusk_blindbidproof_amd/csrc/prover.hip (+ keccak.h for
/// the rng, setup.hip for the generators).
fn cmd_trace(path: &str) {
    let mut failed = 0;
    for line in fs::read_to_string(path).expect("trace file").lines() {
        let f: Vec<&str> = line.split_whitespace().collect();
        if f.len() < 14 || f[0] != "trace" {
            continue;
        }
        let (name, n, toggle) = (f[1], f[2].parse::<usize>().unwrap(), f[3].parse::<u64>().unwrap());
        let sc = |i: usize| Scalar::from_canonical_bytes(arr32(&unhex(f[i]))).expect("canonical scalar in a golden vector");
        let (d, kk, y, y_inv, q, z_img, seed) = (sc(4), sc(5), sc(6), sc(7), sc(8), sc(9), sc(10));
        let bids: Vec<Bid> = (0..n).map(|i| Bid { x: Scalar::from_bits(arr32(&unhex(f[11 + i]))) }).collect();
        let entropy = unhex(f[11 + n]);
        let golden = unhex(f[12 + n]);
        assert_eq!(entropy.len(), 32 * (4 + n) + 32, "entropy = (4 + N) blindings || rng seed");
        let mut script = Vec::new();
        for b in 0..4 + n {
            script.extend_from_slice(&entropy[32 * b..32 * b + 32]);
            script.extend_from_slice(&[0u8; 32]);
        }
        script.extend_from_slice(&entropy[32 * (4 + n)..]);
        // read by the patched rand::thread_rng (rand_patch/README.md); every call of THIS process consumes from the front
        env::set_var("BBP_SCRIPTED_ENTROPY", hex(&script));
        env::set_var("BBP_SCRIPTED_ENTROPY_RESET", name);
        let proof = Proof::prove(d, kk, y, y_inv, q, z_img, seed, bids, toggle).expect("prove");
        let mut record = proof.proof.to_bytes();
        for p in proof.commitments.iter().chain(proof.t_c.iter()) {
            record.extend_from_slice(p.as_bytes());
        }
        if env::var("BBP_SCRIPTED_ENTROPY_USED").is_err() {
            eprintln!("{}: rand::thread_rng is NOT the scripted one (BBP_SCRIPTED_ENTROPY_USED unset): apply rand_patch/README.md first", name);
            std::process::exit(2);
        }
        if record == golden {
            println!("{} IDENTICAL ({} bytes): prover parity pinned for N = {}", name, record.len(), n);
            continue;
        }
        failed += 1;
        println!("{} DIFFERENT (reference {} bytes, golden {} bytes)", name, record.len(), golden.len());
        let plen_ref = record.len() - 32 * (4 + n);
        let plen_gold = golden.len() - 32 * (4 + n);
        // commitments first: they are absorbed before anything else (V_i = v_i B + blinding_i B_blinding)
        let mut first: Option<String> = None;
        for i in 0..4 + n {
            let (a, b) = (&record[plen_ref + 32 * i..plen_ref + 32 * i + 32], &golden[plen_gold + 32 * i..plen_gold + 32 * i + 32]);
            if a != b && first.is_none() {
                first = Some(format!("commitment V_{} (Pedersen generators B / B_blinding, the blinding script, or the committed value)", i));
            }
        }
        if plen_ref != plen_gold {
            println!("    proof part: reference {} bytes, this repository {} bytes: R1CSProof::to_bytes layout (SURVEY.md A.8)", plen_ref, plen_gold);
        }
        let k = 11;
        let mut off = 0;
        for (fname, len, why) in proof_fields(k) {
            if off + len > plen_ref.min(plen_gold) {
                break;
            }
            let same = record[off..off + len] == golden[off..off + len];
            println!("    {:<14} {}", fname, if same { "same".to_string() } else { format!("DIFFERENT  reference {}  golden {}", hex(&record[off..off + len]), hex(&golden[off..off + len])) });
            if !same && first.is_none() {
                first = Some(format!("{}: {}", fname, why));
            }
            off += len;
        }
        println!("    FIRST DIVERGENCE: {}", first.unwrap_or_else(|| "none inside the common prefix: trailing bytes differ".into()));
        let names = ["y", "z", "u", "x", "w"];
        println!("    this repository's challenges for this vector (golden trace), in transcript order:");
        for (i, nm) in names.iter().enumerate() {
            if let Some(v) = f.get(13 + n + i) {
                println!("      {:<4} {}", nm, v);
            }
        }
        for j in 0..k {
            if let Some(v) = f.get(13 + n + names.len() + j) {
                println!("      u_{:<2} {}", j + 1, v);
            }
        }
    }
    if failed != 0 {
        eprintln!("{} vector(s) differ: README.md 'trace' lists the one function per implementation to edit for each step", failed);
        std::process::exit(1);
    }
}

fn main() {
    let a: Vec<String> = env::args().collect();
    match a.get(1).map(|s| s.as_str()) {
        Some("verify") if a.len() == 3 => cmd_verify(&a[2]),
        Some("make") if a.len() == 5 => cmd_make(a[2].parse().unwrap(), a[3].parse().unwrap(), &a[4]),
        Some("layout") => cmd_layout(),
        Some("frames") if a.len() == 3 => cmd_frames(&a[2]),
        Some("trace") if a.len() == 3 => cmd_trace(&a[2]),
        _ => eprintln!("usage: bbp-ref-crosscheck verify VECTORS | make K N OUT | layout | frames FRAMES | trace TRACE"),
    }
}
