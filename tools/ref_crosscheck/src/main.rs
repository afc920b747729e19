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

fn main() {
    let a: Vec<String> = env::args().collect();
    match a.get(1).map(|s| s.as_str()) {
        Some("verify") if a.len() == 3 => cmd_verify(&a[2]),
        Some("make") if a.len() == 5 => cmd_make(a[2].parse().unwrap(), a[3].parse().unwrap(), &a[4]),
        Some("layout") => cmd_layout(),
        Some("frames") if a.len() == 3 => cmd_frames(&a[2]),
        _ => eprintln!("usage: bbp-ref-crosscheck verify VECTORS | make K N OUT | layout | frames FRAMES"),
    }
}
