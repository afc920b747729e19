//! bbp-ref-crosscheck: runs the REFERENCE crate against this repository's golden vectors, and makes reference-made vectors
//! for this repository's verifier.  Written for this repository (it is not a copy of anything in the reference); it only calls
//! the reference's public surface: `Proof::prove` (src/blindbid/proof.rs:36-46), `Verify::new(..).verify()`
//! (src/blindbid/verify.rs:27-89), `Bid` (src/blindbid/bid.rs).
//!
//!   bbp-ref-crosscheck verify  vectors.txt            -> one line per vector: name, accept|reject|format-error
//!   bbp-ref-crosscheck make    K N out.txt             -> K reference-made proofs with an N-entry bid list, same line format
//!   bbp-ref-crosscheck layout                          -> R1CSProof::to_bytes length and first byte of a fresh proof
//!
//! Line format (what tools/ref_crosscheck/export_vectors.py writes and check_reference_made.py reads), all hex, space separated:
//!   name N toggle record score z_img seed pub_0 .. pub_{N-1} [d k y y_inv]
//! `record` = R1CSProof::to_bytes() || 4 commitments || N toggle commitments (include/bbp.h).
use std::env;
use std::fs;
use std::io::Write;

use bulletproofs::r1cs::R1CSProof;
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use dusk_blindbidproof::{Bid, Proof, Verify};
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

fn main() {
    let a: Vec<String> = env::args().collect();
    match a.get(1).map(|s| s.as_str()) {
        Some("verify") if a.len() == 3 => cmd_verify(&a[2]),
        Some("make") if a.len() == 5 => cmd_make(a[2].parse().unwrap(), a[3].parse().unwrap(), &a[4]),
        Some("layout") => cmd_layout(),
        _ => eprintln!("usage: bbp-ref-crosscheck verify VECTORS | make K N OUT | layout"),
    }
}
