// TEST INFRASTRUCTURE: drives the server's request parsers (dusk_blindbidproof_amd/server/wire.h, tlv.h) over hex-encoded request
// bodies read from stdin, one per line ("P <hex>" = opcode-1 body, "V <hex>" = opcode-2 body), and prints what was parsed -- so a
// Python test can compare with its own restatement (tests/uds_client.py) and throw mutated garbage at the parsers.  Built with
// -fsanitize=address,undefined: an out-of-bounds read on a malformed frame fails the test.
#include <stdio.h>

#include <iostream>
#include <string>

#include "../dusk_blindbidproof_amd/server/wire.h"

using namespace bbp_server;

static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) {
        s.push_back(d[p[i] >> 4]);
        s.push_back(d[p[i] & 15]);
    }
    return s;
}

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.size() < 2) continue;
        const char kind = line[0];
        tlv::Bytes body;
        for (size_t i = 2; i + 1 < line.size(); i += 2) body.push_back((uint8_t)std::stoi(line.substr(i, 2), nullptr, 16));
        std::string why;
        if (kind == 'P') {
            ProveRequest r;
            if (parse_prove_request(body.data(), body.size(), &r, &why))
                printf("ok %u %llu %s %s\n", r.n_items, (unsigned long long)r.toggle, hex(r.scalars7, 224).c_str(), hex(r.pub_list.data(), r.pub_list.size()).c_str());
            else
                printf("err %s\n", why.c_str());
        } else {
            VerifyRequest r;
            if (parse_verify_request(body.data(), body.size(), &r, &why))
                printf("ok %u %s %s %s %s %s\n", r.n_items, hex(r.record.data(), r.record.size()).c_str(), hex(r.score, 32).c_str(), hex(r.z_img, 32).c_str(),
                       hex(r.seed, 32).c_str(), hex(r.pub_list.data(), r.pub_list.size()).c_str());
            else
                printf("err %s\n", why.c_str());
        }
        // and the response encoder round trip for whatever record-sized prefix the body offers
        if (body.size() >= 1121 + 32 * 5) {
            const tlv::Bytes f = frame(encode_proof(body.data(), 1121, 1));
            uint64_t len = 0;
            const size_t h = tlv::parse_header(f.data(), f.size(), &len);
            if (!h || h + len != f.size()) printf("BROKEN FRAME\n");
        }
    }
    return 0;
}
