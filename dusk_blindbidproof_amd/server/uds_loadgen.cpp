// bbp-uds-loadgen: closed-loop load generator for bbp-uds-server -- the Go client's role (BenchmarkProveVerify, Readme.md:37-40:
// one prove followed by one verify per op) played by C connections at once, so that the streaming configuration of
// BASELINE.json (configs[4]) can be measured THROUGH the socket.
//
//   bbp-uds-loadgen --socket PATH --requests FILE --connections C --ops M [--no-verify] [--reconnect]
//
// FILE (written by tools/uds_bench.py) holds K pre-encoded bids: u32 len || opcode-1 request frame, u32 len || verify tail
// (the elements that follow the proof blob in an opcode-2 request: score, z_img, seed, public list).  Each connection loops:
// send the prove frame, read the proof frame, wrap it into an opcode-2 request with the bid's tail, expect [0x01].
// Prints one JSON line: ops/s (one op = prove + verify), proofs/s, latency percentiles of prove, verify and the whole op.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "tlv.h"

using namespace bbp_server;
using Clock = std::chrono::steady_clock;

struct Bid {
    tlv::Bytes prove_frame, verify_tail;
};

static bool read_exact(int fd, uint8_t* p, size_t n) {
    while (n) {
        ssize_t r = read(fd, p, n);
        if (r <= 0) {
            if (r < 0 && errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static bool write_all(int fd, const uint8_t* p, size_t n) {
    while (n) {
        ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static bool read_frame(int fd, tlv::Bytes* payload) {
    uint8_t hdr[9];
    if (!read_exact(fd, hdr, 1)) return false;
    const unsigned w = hdr[0];
    if (w != 1 && w != 2 && w != 4 && w != 8) return false;
    if (!read_exact(fd, hdr + 1, w)) return false;
    uint64_t len;
    if (!tlv::parse_header(hdr, 1 + w, &len) || len > (1u << 20)) return false;
    payload->resize((size_t)len);
    return len == 0 || read_exact(fd, payload->data(), (size_t)len);
}
static int dial(const std::string& path) {
    int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    strncpy(a.sun_path, path.c_str(), sizeof a.sun_path - 1);
    for (int tries = 0; tries < 200; tries++) {
        if (connect(fd, (sockaddr*)&a, sizeof a) == 0) return fd;
        if (errno != EAGAIN && errno != ECONNREFUSED) break;
        usleep(5000);  // listen backlog full while thousands of connections arrive at once
    }
    close(fd);
    return -1;
}

int main(int argc, char** argv) {
    std::string sock, file;
    int conns = 64;
    long ops = 1024;
    bool do_verify = true, reconnect = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--socket" && i + 1 < argc) sock = argv[++i];
        else if (a == "--requests" && i + 1 < argc) file = argv[++i];
        else if (a == "--connections" && i + 1 < argc) conns = atoi(argv[++i]);
        else if (a == "--ops" && i + 1 < argc) ops = atol(argv[++i]);
        else if (a == "--no-verify") do_verify = false;
        else if (a == "--reconnect") reconnect = true;  // one connection per request, like a client that dials per call
        else {
            fprintf(stderr, "usage: %s --socket PATH --requests FILE [--connections C] [--ops M] [--no-verify] [--reconnect]\n", argv[0]);
            return 2;
        }
    }
    std::vector<Bid> bids;
    {
        FILE* f = fopen(file.c_str(), "rb");
        if (!f) {
            perror("requests file");
            return 1;
        }
        for (;;) {
            uint32_t n;
            if (fread(&n, 4, 1, f) != 1) break;
            Bid b;
            b.prove_frame.resize(n);
            if (fread(b.prove_frame.data(), 1, n, f) != n || fread(&n, 4, 1, f) != 1) break;
            b.verify_tail.resize(n);
            if (fread(b.verify_tail.data(), 1, n, f) != n) break;
            bids.push_back(std::move(b));
        }
        fclose(f);
    }
    if (bids.empty() || sock.empty()) {
        fprintf(stderr, "no bids loaded or no socket given\n");
        return 1;
    }
    std::atomic<long> next{0}, failed{0}, rejected{0};
    std::vector<std::vector<float>> lat_p(conns), lat_v(conns), lat_o(conns);
    auto worker = [&](int t) {
        int fd = reconnect ? -1 : dial(sock);
        if (!reconnect && fd < 0) {
            failed++;
            return;
        }
        tlv::Bytes proof, reply, body, frame;
        for (;;) {
            const long k = next.fetch_add(1);
            if (k >= ops) break;
            const Bid& b = bids[(size_t)k % bids.size()];
            const auto t0 = Clock::now();
            if (reconnect && (fd = dial(sock)) < 0) {
                failed++;
                break;
            }
            if (!write_all(fd, b.prove_frame.data(), b.prove_frame.size()) || !read_frame(fd, &proof)) {
                failed++;
                break;
            }
            const auto t1 = Clock::now();
            lat_p[t].push_back(std::chrono::duration<float, std::milli>(t1 - t0).count());
            if (do_verify) {
                if (reconnect) {
                    close(fd);
                    if ((fd = dial(sock)) < 0) {
                        failed++;
                        break;
                    }
                }
                body.assign(1, 0x02);
                tlv::write(body, proof);
                body.insert(body.end(), b.verify_tail.begin(), b.verify_tail.end());
                frame.clear();
                tlv::write(frame, body);
                if (!write_all(fd, frame.data(), frame.size()) || !read_frame(fd, &reply)) {
                    failed++;
                    break;
                }
                if (reply.size() != 1 || reply[0] != 0x01) rejected++;
                const auto t2 = Clock::now();
                lat_v[t].push_back(std::chrono::duration<float, std::milli>(t2 - t1).count());
                lat_o[t].push_back(std::chrono::duration<float, std::milli>(t2 - t0).count());
            }
            if (reconnect) close(fd);
        }
        if (!reconnect && fd >= 0) close(fd);
    };
    const auto T0 = Clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < conns; t++) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    const double wall = std::chrono::duration<double>(Clock::now() - T0).count();
    auto pct = [](std::vector<std::vector<float>>& v, std::vector<float>* all) {
        all->clear();
        for (auto& x : v) all->insert(all->end(), x.begin(), x.end());
        std::sort(all->begin(), all->end());
    };
    auto at = [](const std::vector<float>& a, double q) { return a.empty() ? 0.f : a[std::min(a.size() - 1, (size_t)(a.size() * q))]; };
    std::vector<float> p, v, o;
    pct(lat_p, &p);
    pct(lat_v, &v);
    pct(lat_o, &o);
    printf("{\"connections\": %d, \"ops\": %zu, \"wall_s\": %.3f, \"proofs_per_s\": %.1f, \"verifies_per_s\": %.1f, \"failed\": %ld, \"rejected\": %ld, "
           "\"prove_latency_ms\": {\"p50\": %.2f, \"p99\": %.2f}, \"verify_latency_ms\": {\"p50\": %.2f, \"p99\": %.2f}, "
           "\"op_latency_ms\": {\"p50\": %.2f, \"p99\": %.2f}, \"reconnect\": %s}\n",
           conns, p.size(), wall, p.size() / wall, v.size() / wall, failed.load(), rejected.load(), at(p, 0.5), at(p, 0.99), at(v, 0.5), at(v, 0.99),
           at(o, 0.5), at(o, 0.99), reconnect ? "true" : "false");
    return failed.load() || rejected.load() ? 1 : 0;
}
