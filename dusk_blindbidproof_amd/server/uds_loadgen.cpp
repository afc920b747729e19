// bbp-uds-loadgen: load generator for bbp-uds-server -- the Go client's role (BenchmarkProveVerify, Readme.md:37-40: one prove
// followed by one verify per op) played by many connections at once, so that the streaming configuration of BASELINE.json
// (configs[4]) can be measured THROUGH the socket.  Event-driven (epoll, a few threads), so that the generator does not spend the
// host's cores on thousands of threads the server then competes with.
//
//   closed loop:  bbp-uds-loadgen --socket PATH --requests FILE --connections C --ops M [--no-verify]
//                 C connections, each runs one op after the other until M ops are done: throughput at a given concurrency.
//   open loop:    bbp-uds-loadgen --socket PATH --requests FILE --rate R --duration S [--connections CAP] [--no-verify]
//                 ops ARRIVE as a Poisson process of R per second whatever the server does (the load a network of bidders puts
//                 on a node); an arrival takes an idle connection, opens a new one (up to CAP), or waits in the generator's
//                 backlog.  Latencies are measured from the SCHEDULED arrival, so a server that falls behind shows it as
//                 latency instead of hiding it (no coordinated omission).
//
// FILE (written by tools/uds_bench.py) holds K pre-encoded bids: u32 len || opcode-1 request frame, u32 len || verify tail
// (the elements that follow the proof blob in an opcode-2 request: score, z_img, seed, public list).  One op: send the prove
// frame, read the proof frame, wrap it into an opcode-2 request with the bid's tail, expect [0x01].
// Prints one JSON line: ops/s, proofs/s, latency percentiles of prove, verify and the whole op.
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/epoll.h>
#include <sys/resource.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "tlv.h"

using namespace bbp_server;
using Clock = std::chrono::steady_clock;

struct Bid {
    tlv::Bytes prove_frame, verify_tail;
};

static bool write_all(int fd, const uint8_t* p, size_t n) {  // blocking send: requests are a few KB, far below the socket buffer
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
static std::atomic<long> g_dial_retries{0}, g_dial_us{0};
static int dial_inner(const std::string& path);
static int dial(const std::string& path) {
    const auto t0 = Clock::now();
    const int fd = dial_inner(path);
    g_dial_us += std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count();
    return fd;
}
static int dial_inner(const std::string& path) {
    int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0) return -1;
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    strncpy(a.sun_path, path.c_str(), sizeof a.sun_path - 1);
    for (int tries = 0; tries < 400; tries++) {
        if (connect(fd, (sockaddr*)&a, sizeof a) == 0) return fd;
        if (errno != EAGAIN && errno != ECONNREFUSED) break;
        g_dial_retries++;
        usleep(2000);  // listen backlog full while thousands of connections arrive at once
    }
    close(fd);
    return -1;
}

enum State { IDLE, WAIT_PROOF, WAIT_VERIFY };
struct Conn {
    int fd = -1;
    State st = IDLE;
    size_t bid = 0;
    Clock::time_point t_sched, t_sent, t_proof;
    tlv::Bytes in;
};

struct Shared {
    std::string sock;
    std::vector<Bid> bids;
    bool do_verify = true;
    std::atomic<long> next{0}, failed{0}, rejected{0};
    long ops = 0;  // closed loop
};

struct Worker {
    Shared* sh = nullptr;
    int ep = -1;
    std::vector<Conn> conns;
    std::vector<int> idle;  // indices into conns
    std::vector<float> lat_p, lat_v, lat_o;
    std::vector<double> t_done;  // completion time of every op, seconds on the steady clock
    // open loop
    double rate = 0;
    double duration = 0;
    int cap = 0;
    int preconnect = 0;  // open loop: connections opened before the clock starts (a population of clients with persistent connections)
    std::deque<Clock::time_point> backlog;
    size_t max_backlog = 0;
    long arrivals = 0;
    int active = 0;

    bool add_conn() {
        const int fd = dial(sh->sock);
        if (fd < 0) return false;
        Conn c;
        c.fd = fd;
        conns.push_back(std::move(c));
        epoll_event ev;
        memset(&ev, 0, sizeof ev);
        ev.events = EPOLLIN;
        ev.data.u32 = (uint32_t)(conns.size() - 1);
        epoll_ctl(ep, EPOLL_CTL_ADD, fd, &ev);
        idle.push_back((int)conns.size() - 1);
        return true;
    }
    void fail(Conn& c) {
        sh->failed++;
        if (c.fd >= 0) {
            epoll_ctl(ep, EPOLL_CTL_DEL, c.fd, nullptr);
            close(c.fd);
            c.fd = -1;
        }
        if (c.st != IDLE) active--;
        c.st = IDLE;  // never reused: not in `idle`
    }
    void start(int ci, long k, Clock::time_point t_sched) {
        Conn& c = conns[(size_t)ci];
        c.bid = (size_t)k % sh->bids.size();
        c.t_sched = t_sched;
        c.t_sent = Clock::now();
        c.st = WAIT_PROOF;
        c.in.clear();
        active++;
        const Bid& b = sh->bids[c.bid];
        if (!write_all(c.fd, b.prove_frame.data(), b.prove_frame.size())) fail(c);
    }
    // a whole frame in c.in?  -> payload
    static int take_frame(Conn& c, tlv::Bytes* payload) {
        if (!tlv::header_complete(c.in.data(), c.in.size())) return 0;
        uint64_t len;
        const size_t h = tlv::parse_header(c.in.data(), c.in.size(), &len);
        if (!h || len > (1u << 20)) return -1;
        if (c.in.size() - h < len) return 0;
        payload->assign(c.in.begin() + (ptrdiff_t)h, c.in.begin() + (ptrdiff_t)(h + len));
        c.in.erase(c.in.begin(), c.in.begin() + (ptrdiff_t)(h + len));
        return 1;
    }
    void op_done(int ci) {
        Conn& c = conns[(size_t)ci];
        t_done.push_back(std::chrono::duration<double>(Clock::now().time_since_epoch()).count());
        c.st = IDLE;
        active--;
        idle.push_back(ci);
    }
    void on_readable(int ci) {
        Conn& c = conns[(size_t)ci];
        uint8_t buf[8192];
        for (;;) {
            const ssize_t r = recv(c.fd, buf, sizeof buf, MSG_DONTWAIT);
            if (r > 0) {
                c.in.insert(c.in.end(), buf, buf + r);
                if ((size_t)r < sizeof buf) break;
                continue;
            }
            if (r < 0 && errno == EINTR) continue;
            if (r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) break;
            if (c.st != IDLE || r < 0) fail(c);  // closed under a request (prove error: the server writes nothing and drops the connection)
            else {
                epoll_ctl(ep, EPOLL_CTL_DEL, c.fd, nullptr);
                close(c.fd);
                c.fd = -1;
                idle.erase(std::remove(idle.begin(), idle.end(), ci), idle.end());
            }
            return;
        }
        tlv::Bytes payload;
        const int got = take_frame(c, &payload);
        if (got < 0) return fail(c);
        if (got == 0) return;
        const auto now = Clock::now();
        const auto base = rate > 0 ? c.t_sched : c.t_sent;
        if (c.st == WAIT_PROOF) {
            lat_p.push_back(std::chrono::duration<float, std::milli>(now - base).count());
            c.t_proof = now;
            if (!sh->do_verify) return op_done(ci);
            tlv::Bytes body(1, 0x02), frame;
            tlv::write(body, payload);
            const Bid& b = sh->bids[c.bid];
            body.insert(body.end(), b.verify_tail.begin(), b.verify_tail.end());
            tlv::write(frame, body);
            c.st = WAIT_VERIFY;
            if (!write_all(c.fd, frame.data(), frame.size())) fail(c);
        } else if (c.st == WAIT_VERIFY) {
            if (payload.size() != 1 || payload[0] != 0x01) sh->rejected++;
            lat_v.push_back(std::chrono::duration<float, std::milli>(now - c.t_proof).count());
            lat_o.push_back(std::chrono::duration<float, std::milli>(now - base).count());
            op_done(ci);
        }
    }
    int take_idle() {
        while (!idle.empty()) {
            const int ci = idle.back();
            idle.pop_back();
            if (conns[(size_t)ci].fd >= 0) return ci;
        }
        return -1;
    }
    void pump(int timeout_ms) {
        epoll_event evs[256];
        const int n = epoll_wait(ep, evs, 256, timeout_ms);
        for (int i = 0; i < n; i++) on_readable((int)evs[i].data.u32);
    }
    void run_closed(int n_conns) {
        for (int i = 0; i < n_conns; i++)
            if (!add_conn()) sh->failed++;
        for (;;) {
            for (;;) {  // every idle connection takes the next op while there are ops left
                if (idle.empty()) break;
                const long k = sh->next.fetch_add(1);
                if (k >= sh->ops) break;
                const int ci = take_idle();
                if (ci < 0) {
                    sh->next.fetch_sub(1);
                    break;
                }
                start(ci, k, Clock::now());
            }
            if (active == 0 && (sh->next.load() >= sh->ops || idle.empty())) break;
            pump(100);
        }
    }
    void run_open(unsigned seed) {
        std::mt19937_64 rng(seed);
        std::exponential_distribution<double> gap(rate);
        for (int i = 0; i < preconnect && i < cap; i++)
            if (!add_conn()) break;
        const auto T0 = Clock::now();
        auto next_arrival = T0 + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(gap(rng)));
        const auto T_end = T0 + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(duration));
        bool arriving = true;
        for (;;) {
            auto now = Clock::now();
            while (arriving && next_arrival <= now) {
                if (next_arrival >= T_end) {
                    arriving = false;
                    break;
                }
                backlog.push_back(next_arrival);
                arrivals++;
                next_arrival += std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(gap(rng)));
            }
            while (!backlog.empty()) {  // serve the backlog in arrival order with whatever connections are free / may still be opened
                int ci = take_idle();
                if (ci < 0 && (int)conns.size() < cap && add_conn()) ci = take_idle();
                if (ci < 0) break;
                start(ci, sh->next.fetch_add(1), backlog.front());
                backlog.pop_front();
            }
            if (backlog.size() > max_backlog) max_backlog = backlog.size();
            if (!arriving && backlog.empty() && active == 0) break;
            if (!arriving && Clock::now() > T_end + std::chrono::seconds(30)) break;  // a server that never answers
            int wait_ms = 50;
            if (arriving) {
                const auto d = std::chrono::duration_cast<std::chrono::microseconds>(next_arrival - Clock::now()).count();
                wait_ms = d <= 0 ? 0 : (int)std::min<long>(50, (d + 999) / 1000);
            }
            pump(wait_ms);
        }
    }
};

int main(int argc, char** argv) {
    Shared sh;
    std::string file;
    int conns = 64, threads = 2, preconnect = 0;
    double rate = 0, duration = 10;
    sh.ops = 1024;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--socket" && i + 1 < argc) sh.sock = argv[++i];
        else if (a == "--requests" && i + 1 < argc) file = argv[++i];
        else if (a == "--connections" && i + 1 < argc) conns = atoi(argv[++i]);
        else if (a == "--ops" && i + 1 < argc) sh.ops = atol(argv[++i]);
        else if (a == "--rate" && i + 1 < argc) rate = atof(argv[++i]);
        else if (a == "--duration" && i + 1 < argc) duration = atof(argv[++i]);
        else if (a == "--threads" && i + 1 < argc) threads = atoi(argv[++i]);
        else if (a == "--preconnect" && i + 1 < argc) preconnect = atoi(argv[++i]);
        else if (a == "--no-verify") sh.do_verify = false;
        else {
            fprintf(stderr,
                    "usage: %s --socket PATH --requests FILE [--connections C] [--ops M | --rate R --duration S] [--threads T] [--no-verify]\n", argv[0]);
            return 2;
        }
    }
    {
        rlimit rl;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < rl.rlim_max) {
            rl.rlim_cur = rl.rlim_max;
            setrlimit(RLIMIT_NOFILE, &rl);
        }
    }
    {  // the descriptor table at its final size before the threads start (a doubling under threads waits for an RCU grace period: ~0.1 s)
        const int hi = fcntl(0, F_DUPFD_CLOEXEC, conns + 64);
        if (hi >= 0) close(hi);
    }
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
            sh.bids.push_back(std::move(b));
        }
        fclose(f);
    }
    if (sh.bids.empty() || sh.sock.empty()) {
        fprintf(stderr, "no bids loaded or no socket given\n");
        return 1;
    }
    if (threads < 1) threads = 1;
    if (threads > conns) threads = conns;
    std::vector<Worker> ws((size_t)threads);
    for (int t = 0; t < threads; t++) {
        ws[(size_t)t].sh = &sh;
        ws[(size_t)t].ep = epoll_create1(EPOLL_CLOEXEC);
        ws[(size_t)t].rate = rate / threads;
        ws[(size_t)t].duration = duration;
        ws[(size_t)t].cap = conns / threads + (t < conns % threads ? 1 : 0);
        ws[(size_t)t].preconnect = preconnect / threads;
    }
    const auto T0 = Clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            if (rate > 0) ws[(size_t)t].run_open(12345u + (unsigned)t);
            else ws[(size_t)t].run_closed(ws[(size_t)t].cap);
        });
    for (auto& x : th) x.join();
    const double wall = std::chrono::duration<double>(Clock::now() - T0).count();
    std::vector<float> p, v, o;
    std::vector<double> td;
    size_t max_backlog = 0, n_conns = 0;
    long arrivals = 0;
    for (auto& w : ws) {
        p.insert(p.end(), w.lat_p.begin(), w.lat_p.end());
        v.insert(v.end(), w.lat_v.begin(), w.lat_v.end());
        o.insert(o.end(), w.lat_o.begin(), w.lat_o.end());
        td.insert(td.end(), w.t_done.begin(), w.t_done.end());
        max_backlog += w.max_backlog;
        n_conns += w.conns.size();
        arrivals += w.arrivals;
    }
    std::sort(p.begin(), p.end());
    std::sort(v.begin(), v.end());
    std::sort(o.begin(), o.end());
    // sustained rate: completions between the 10th and the 90th percentile of the run (the ramp -- the first batches form while the
    // connections are still being opened -- and the tail -- the last, partial batches -- are what a short run adds to wall_s)
    std::sort(td.begin(), td.end());
    double sustained = 0;
    if (td.size() >= 100) {
        const size_t a = td.size() / 10, b = td.size() - td.size() / 10 - 1;
        if (td[b] > td[a]) sustained = (double)(b - a) / (td[b] - td[a]);
    }
    auto at = [](const std::vector<float>& a, double q) { return a.empty() ? 0.f : a[std::min(a.size() - 1, (size_t)(a.size() * q))]; };
    printf("{\"mode\": \"%s\", \"connections\": %zu, \"ops\": %zu, \"wall_s\": %.3f, \"proofs_per_s\": %.1f, \"verifies_per_s\": %.1f, \"failed\": %ld, "
           "\"rejected\": %ld, \"prove_latency_ms\": {\"p50\": %.2f, \"p99\": %.2f}, \"verify_latency_ms\": {\"p50\": %.2f, \"p99\": %.2f}, "
           "\"op_latency_ms\": {\"p50\": %.2f, \"p99\": %.2f}",
           rate > 0 ? "open-loop (Poisson arrivals; latency from the scheduled arrival)" : "closed-loop", n_conns, p.size(), wall, p.size() / wall,
           v.size() / wall, sh.failed.load(), sh.rejected.load(), at(p, 0.5), at(p, 0.99), at(v, 0.5), at(v, 0.99), at(o, 0.5), at(o, 0.99));
    if (rate > 0)
        printf(", \"offered_per_s\": %.1f, \"duration_s\": %.1f, \"arrivals\": %ld, \"max_client_backlog\": %zu, \"connection_cap\": %d", rate, duration,
               arrivals, max_backlog, conns);
    printf(", \"sustained_ops_per_s\": %.1f, \"generator_threads\": %d, \"dial_ms_total\": %.1f, \"dial_retries\": %ld}\n", sustained, threads,
           g_dial_us.load() / 1e3, g_dial_retries.load());
    return sh.failed.load() || sh.rejected.load() ? 1 : 0;
}
