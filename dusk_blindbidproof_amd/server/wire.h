// Request / response layouts of the reference's IPC surface, on top of tlv.h (which alone knows the framing bytes).
// Mirrors, element for element:
//   prove request    src/futures/main.rs:70-85 + src/blindbid/proof.rs:97-115   (try_from_reader_variables)
//   prove response   src/blindbid/proof.rs:118-143 (TryInto<Vec<u8>>) wrapped by src/futures/main.rs:89-90
//   verify request   src/futures/main.rs:94 + src/blindbid/verify.rs:91-128 + proof.rs:145-183 (TryFrom<Vec<u8>>)
//   verify response  src/futures/main.rs:95-99
#pragma once
#include "tlv.h"

namespace bbp_server {

constexpr uint8_t OP_PROVE = 1, OP_VERIFY = 2;  // src/futures/main.rs:84,94
constexpr uint32_t MAX_ITEMS = 202;             // 1442 + 3N <= 2048 generators (R1CSError::InvalidGeneratorsLength beyond)

struct ProveRequest {
    uint8_t scalars7[7 * 32];  // d, k, y, y_inv, q, z_img, seed (proof.rs:100-106)
    tlv::Bytes pub_list;       // N x 32 raw bytes; Bid::from -> Scalar::from_bits (bid.rs:20-29) happens in the engine
    uint64_t toggle = 0;       // proof.rs:112
    uint32_t n_items = 0;
};

// body = request[1..] (the opcode byte already consumed, main.rs:85)
inline bool parse_prove_request(const uint8_t* body, size_t len, ProveRequest* out, std::string* why) {
    tlv::Reader r(body, len);
    for (int i = 0; i < 7; i++)
        if (!r.read_32(out->scalars7 + 32 * i)) return *why = "scalar field missing or not 32 bytes", false;
    std::vector<tlv::Bytes> bids;
    if (!r.read_list(&bids)) return *why = "bid list missing or malformed", false;  // Bid::try_list_from_reader (bid.rs:15-17)
    if (bids.empty()) return *why = "empty bid list (the reference panics at src/gadgets.rs:103)", false;
    if (bids.size() > MAX_ITEMS) return *why = "bid list needs more than 2048 multipliers", false;
    out->pub_list.clear();
    for (const tlv::Bytes& b : bids) {
        // Bid::from(Vec<u8>) (bid.rs:20-29): anything but exactly 32 bytes panics there (cmp::max) and the release
        // profile aborts the process; here it is an error reply like every other malformed request
        if (b.size() != 32) return *why = "bid is not 32 bytes", false;
        out->pub_list.insert(out->pub_list.end(), b.begin(), b.end());
    }
    out->n_items = (uint32_t)bids.size();
    if (!r.read_u64(&out->toggle)) return *why = "toggle missing or not a u64", false;
    return true;
}

// record = R1CSProof bytes || 4 x 32 commitments || N x 32 t_c (include/bbp.h) -> Proof::try_into (proof.rs:118-143)
inline tlv::Bytes encode_proof(const uint8_t* record, uint32_t proof_len, uint32_t n_items) {
    tlv::Bytes out;
    tlv::write(out, record, proof_len);  // buf.write(self.proof.to_bytes())
    std::vector<tlv::Bytes> c, t;
    for (uint32_t i = 0; i < 4; i++) c.emplace_back(record + proof_len + 32 * i, record + proof_len + 32 * (i + 1));
    for (uint32_t i = 0; i < n_items; i++) t.emplace_back(record + proof_len + 128 + 32 * i, record + proof_len + 128 + 32 * (i + 1));
    tlv::write_list(out, c);  // buf.write_list(commitments)
    tlv::write_list(out, t);  // buf.write_list(t_c)
    return out;
}

// one frame on the socket: TlvWriter::new(s).write(payload) (main.rs:89-90, 98-99)
inline tlv::Bytes frame(const tlv::Bytes& payload) {
    tlv::Bytes out;
    tlv::write(out, payload);
    return out;
}

struct VerifyRequest {
    tlv::Bytes record;          // R1CSProof bytes || commitments || t_c, as bbp_verify takes it
    uint8_t score[32], z_img[32], seed[32];
    tlv::Bytes pub_list;        // N x 32 raw bytes (Scalar::from_bits in the engine, verify.rs:115)
    uint32_t n_items = 0;
};

// Proof::try_from(Vec<u8>) (proof.rs:145-183): proof element, commitments list, t_c list; every point exactly 32 bytes
inline bool parse_proof_blob(const uint8_t* p, size_t len, tlv::Bytes* record, uint32_t* n_tc, std::string* why) {
    tlv::Reader r(p, len);
    const uint8_t* pr;
    size_t pl;
    if (!r.next(&pr, &pl)) return *why = "The proof was not supplied", false;
    std::vector<tlv::Bytes> c, t;
    if (!r.read_list(&c) || !r.read_list(&t)) return *why = "commitment lists missing or malformed", false;
    for (const auto& v : {&c, &t})
        for (const tlv::Bytes& b : *v)
            if (b.size() != 32) return *why = "Compressed Ristrettos can only be created from 32 bytes slices", false;
    // Verify::verify indexes vars[0], vars[1], vars[3] (verify.rs:76-78): fewer than four commitments panic there; more than four
    // change the transcript ("V" appended per commitment) in a way no honest prover produces -- rejected here, not aborted
    if (c.size() != 4) return *why = "exactly four commitments expected", false;
    if (t.empty() || t.size() > MAX_ITEMS) return *why = "toggle commitment count out of range", false;
    record->assign(pr, pr + pl);
    for (const tlv::Bytes& b : c) record->insert(record->end(), b.begin(), b.end());
    for (const tlv::Bytes& b : t) record->insert(record->end(), b.begin(), b.end());
    *n_tc = (uint32_t)t.size();
    return true;
}

// body = request[1..]: verify.rs:91-128
inline bool parse_verify_request(const uint8_t* body, size_t len, VerifyRequest* out, std::string* why) {
    tlv::Reader r(body, len);
    const uint8_t* blob;
    size_t bl;
    if (!r.next(&blob, &bl)) return *why = "No proof data was provided", false;
    uint32_t n_tc = 0;
    if (!parse_proof_blob(blob, bl, &out->record, &n_tc, why)) return false;
    if (!r.read_32(out->score) || !r.read_32(out->z_img) || !r.read_32(out->seed)) return *why = "public scalar missing or not 32 bytes", false;
    std::vector<tlv::Bytes> items;
    if (!r.read_list(&items)) return *why = "public list missing or malformed", false;
    for (const tlv::Bytes& b : items)
        if (b.size() != 32) return *why = "Scalars Ristrettos can only be created from 32 bytes slices", false;
    // one_of_many_gadget walks toggle.len() entries of items (gadgets.rs:97-131): a shorter public list panics in the reference,
    // extra entries are never read
    if (items.size() < n_tc) return *why = "public list shorter than the toggle commitments", false;
    out->pub_list.clear();
    for (uint32_t i = 0; i < n_tc; i++) out->pub_list.insert(out->pub_list.end(), items[i].begin(), items[i].end());
    out->n_items = n_tc;
    return true;
}

}  // namespace bbp_server
