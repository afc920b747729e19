// dusk-tlv framing -- THE ONE FILE that knows the byte layout.            *** PARITY UNPINNED ***
//
// The reference frames every IPC message with the `dusk-tlv` crate (Cargo.toml:21, git 5be856b / v1.0.1; call sites
// src/futures/main.rs:70-99, src/blindbid/proof.rs:97-170, verify.rs:91-105, bid.rs:14-18).  Its source is NOT in the reference
// snapshot and cannot be fetched, so the byte layout below is restated from the crate's public behaviour as
// remembered, and nothing in /root/reference pins it.  Everything else in server/ goes through Reader / Writer, so a maintainer
// with the crate at hand corrects THIS file only.  What is assumed, byte by byte:
//
//   element   := width(1 byte) || length(width bytes, little-endian) || payload(length bytes)
//                width is the number of bytes the length field takes: 1, 2, 4 or 8 -- the writer picks the smallest that holds
//                the length (TlvWriter::write);  [unknown: whether the crate's tag is this width or some other type code]
//   list      := element whose payload is the concatenation of the items as elements (TlvWriter::write_list / TlvReader::read_list)
//   Scalar    := element of 32 bytes -- curve25519-dalek 1.x serialises a Scalar with serde `serialize_bytes`, and dusk-tlv's serde
//                Deserializer hands the next element's payload to the visitor (proof.rs:100-106); must be canonical (< l)
//   u64       := element of 8 bytes, little-endian (proof.rs:112)               [unknown: could be a minimal-width integer]
//
// What IS pinned by the reference's own code: the ORDER and nesting of elements (wire.h cites each line) and the opcode byte.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace bbp_server {
namespace tlv {

using Bytes = std::vector<uint8_t>;

// appends element(payload) to out
inline void write(Bytes& out, const uint8_t* payload, size_t len) {
    const int width = len <= 0xffu ? 1 : len <= 0xffffu ? 2 : len <= 0xffffffffu ? 4 : 8;
    out.push_back((uint8_t)width);
    for (int i = 0; i < width; i++) out.push_back((uint8_t)((uint64_t)len >> (8 * i)));
    out.insert(out.end(), payload, payload + len);
}
inline void write(Bytes& out, const Bytes& payload) { write(out, payload.data(), payload.size()); }

// appends list(items) -- TlvWriter::write_list
inline void write_list(Bytes& out, const std::vector<Bytes>& items) {
    Bytes inner;
    for (const Bytes& it : items) write(inner, it);
    write(out, inner);
}

// header of the element that starts at p: returns header size (0 = malformed / truncated header) and the payload length
inline size_t parse_header(const uint8_t* p, size_t avail, uint64_t* len) {
    if (avail < 1) return 0;
    const unsigned width = p[0];
    if (width != 1 && width != 2 && width != 4 && width != 8) return 0;
    if (avail < 1 + width) return 0;
    uint64_t v = 0;
    for (unsigned i = 0; i < width; i++) v |= (uint64_t)p[1 + i] << (8 * i);
    *len = v;
    return 1 + width;
}
inline bool header_complete(const uint8_t* p, size_t avail) {  // enough bytes to know the element's total size?
    if (avail < 1) return false;
    const unsigned width = p[0];
    if (width != 1 && width != 2 && width != 4 && width != 8) return true;  // malformed: parse_header will say so
    return avail >= 1 + width;
}

// sequential reader over a byte span -- TlvReader over &[u8]
class Reader {
  public:
    Reader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    bool at_end() const { return n_ == 0; }
    // next element's payload (TlvReader::next): false on end of input or malformed / truncated element
    bool next(const uint8_t** payload, size_t* len) {
        uint64_t l;
        const size_t h = parse_header(p_, n_, &l);
        if (!h || l > n_ - h) return false;
        *payload = p_ + h;
        *len = (size_t)l;
        p_ += h + l;
        n_ -= h + (size_t)l;
        return true;
    }
    // TlvReader::read_list::<Vec<u8>>: an element whose payload is a sequence of elements
    bool read_list(std::vector<Bytes>* items) {
        const uint8_t* q;
        size_t l;
        if (!next(&q, &l)) return false;
        Reader inner(q, l);
        items->clear();
        while (!inner.at_end()) {
            const uint8_t* it;
            size_t il;
            if (!inner.next(&it, &il)) return false;
            items->emplace_back(it, it + il);
        }
        return true;
    }
    // serde Deserialize of a Scalar through the reader: 32 bytes
    bool read_32(uint8_t out[32]) {
        const uint8_t* q;
        size_t l;
        if (!next(&q, &l) || l != 32) return false;
        memcpy(out, q, 32);
        return true;
    }
    bool read_u64(uint64_t* v) {
        const uint8_t* q;
        size_t l;
        if (!next(&q, &l) || l != 8) return false;
        uint64_t x = 0;
        for (int i = 0; i < 8; i++) x |= (uint64_t)q[i] << (8 * i);
        *v = x;
        return true;
    }
    const uint8_t* rest() const { return p_; }
    size_t rest_len() const { return n_; }

  private:
    const uint8_t* p_;
    size_t n_;
};

}  // namespace tlv
}  // namespace bbp_server
