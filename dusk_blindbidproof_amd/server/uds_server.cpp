// bbp-uds-server: the reference's IPC surface in front of the MI355X engine.
//
// Replaces src/main.rs (CLI, bind) + src/futures/{main,prove,verify}.rs (one TLV frame per request, opcode byte 1 = prove,
// 2 = verify) + the dusk-uds worker pool.  Same observable behaviour on the socket:
//   * prove  (opcode 1): reply frame = TLV(proof) || LIST(4 commitments) || LIST(N toggle commitments); on ANY error -- malformed
//     request, non-canonical scalar, engine failure -- NOTHING is written (Message::Error, src/futures/main.rs:15-25,86-92)
//   * verify (opcode 2): reply frame = [0x01] on accept, [0x00] on reject OR malformed request (main.rs:94-101: parse errors
//     collapse into `.is_ok() == false`)
//   * unknown opcode / unreadable frame: nothing written (main.rs:102-105)
// and the same CLI flags (-b/--bind-path, default $TMPDIR/dusk-uds-blindbid; -l/--log-level, RUST_LOG wins when set:
// src/main.rs:14-48).  The reference never aborts a connection's peer on a panic-worthy request because it aborts itself
// (panic = 'abort', Cargo.toml:29); this server answers such requests like any other malformed one.
//
// What is different by design: the reference runs each request on its own worker thread, N requests = N CPU proofs side by
// side.  Here each connection also gets a thread, but the threads only parse and wait: bbp_prove / bbp_verify hand the request
// to the engine's call combiner, which turns whatever is in flight into ONE device batch (--window-us: how long a batch leader
// waits for company).  The engine is loaded with dlopen (--engine), so this binary holds no GPU code and the CPU-tier tests can
// run it against a stub.
#include <dlfcn.h>
#include <errno.h>
#include <signal.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bbp.h"
#include "wire.h"

using namespace bbp_server;

// ---- logging (env_logger stand-in: levels error < warn < info < debug < trace, to stderr) ---------------------------------------
static int g_level = 2;
static int level_of(const char* s) {
    static const char* names[] = {"error", "warn", "info", "debug", "trace"};
    for (int i = 0; i < 5; i++)
        if (strcasecmp(s, names[i]) == 0) return i;
    return -1;
}
static void logf(int lvl, const char* fmt, ...) {
    if (lvl > g_level) return;
    static const char* names[] = {"ERROR", "WARN", "INFO", "DEBUG", "TRACE"};
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "[%s bbp-uds-server] %s\n", names[lvl], buf);
}

// ---- engine (the C ABI of include/bbp.h, bound at run time) ----------------------------------------------------------------------
struct Engine {
    void* so = nullptr;
    // one context per GPU (--devices 0,1,..): bids are independent, so connections are dealt round-robin over the contexts and
    // every context batches its own share (batch-sharded, no cross-GPU traffic at all on this path)
    std::vector<bbp_ctx*> ctxs;
    std::atomic<uint64_t> rr{0};
    bbp_ctx* pick() { return ctxs[rr.fetch_add(1) % ctxs.size()]; }
    decltype(&bbp_init) init = nullptr;
    decltype(&bbp_free) free_ = nullptr;
    decltype(&bbp_prove) prove = nullptr;
    decltype(&bbp_verify) verify = nullptr;
    decltype(&bbp_last_error) last_error = nullptr;
    decltype(&bbp_proof_record_size) record_size = nullptr;
    decltype(&bbp_set_batching) set_batching = nullptr;
    decltype(&bbp_batching_stats) batching_stats = nullptr;
    bool load(const char* path, std::string* why) {
        so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!so) return *why = dlerror(), false;
        auto sym = [&](const char* n) {
            void* p = dlsym(so, n);
            if (!p && why->empty()) *why = std::string("missing symbol ") + n;
            return p;
        };
        init = (decltype(init))sym("bbp_init");
        free_ = (decltype(free_))sym("bbp_free");
        prove = (decltype(prove))sym("bbp_prove");
        verify = (decltype(verify))sym("bbp_verify");
        last_error = (decltype(last_error))sym("bbp_last_error");
        record_size = (decltype(record_size))sym("bbp_proof_record_size");
        set_batching = (decltype(set_batching))sym("bbp_set_batching");
        batching_stats = (decltype(batching_stats))sym("bbp_batching_stats");
        return why->empty();
    }
};
static Engine g_eng;
static std::atomic<uint64_t> g_served{0}, g_errors{0};
static std::atomic<int> g_live{0};
static std::mutex g_live_mu;
static std::condition_variable g_live_cv;
static int g_max_conn = 1024;
static volatile sig_atomic_t g_stop = 0;
static int g_listen_fd = -1;

// ---- socket helpers --------------------------------------------------------------------------------------------------------------
static bool read_exact(int fd, uint8_t* p, size_t n) {
    while (n) {
        ssize_t r = read(fd, p, n);
        if (r == 0) return false;
        if (r < 0) {
            if (errno == EINTR) continue;
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

constexpr uint64_t MAX_FRAME = 1u << 20;  // the largest legitimate request (verify, N = 202) is ~ 16 KB

// TlvReader::new(socket).next() (main.rs:70-79): 1 = frame read, 0 = clean end of stream before a frame, -1 = error
static int read_frame(int fd, tlv::Bytes* payload) {
    uint8_t hdr[9];
    ssize_t r;
    do r = read(fd, hdr, 1);
    while (r < 0 && errno == EINTR);
    if (r == 0) return 0;
    if (r < 0) return -1;
    const unsigned width = hdr[0];
    if (width != 1 && width != 2 && width != 4 && width != 8) return -1;
    if (!read_exact(fd, hdr + 1, width)) return -1;
    uint64_t len;
    if (!tlv::parse_header(hdr, 1 + width, &len) || len > MAX_FRAME) return -1;
    payload->resize((size_t)len);
    if (len && !read_exact(fd, payload->data(), (size_t)len)) return -1;
    return 1;
}

// ---- one connection: frames until the peer closes (a reference-style client sends one and reads one) -----------------------------
static void serve(int fd) {
    bbp_ctx* const ctx = g_eng.pick();  // this connection's GPU
    tlv::Bytes req;
    for (;;) {
        const int got = read_frame(fd, &req);
        if (got == 0) break;
        if (got < 0 || req.empty()) {
            logf(0, "Error resolving the request: %s", got < 0 ? "unreadable frame" : "The request was not provided");
            g_errors++;
            break;  // Message::Error: nothing written
        }
        const uint8_t opcode = req[0];
        std::string why;
        if (opcode == OP_PROVE) {
            ProveRequest pr;
            int32_t rc = BBP_ERR_FORMAT;
            tlv::Bytes record;
            uint32_t proof_len = 0;
            if (parse_prove_request(req.data() + 1, req.size() - 1, &pr, &why)) {
                record.resize(g_eng.record_size(pr.n_items));
                rc = g_eng.prove(ctx, pr.scalars7, pr.pub_list.data(), pr.n_items, pr.toggle, nullptr, record.data(), &proof_len);
                if (rc != BBP_OK) why = std::string("engine status ") + std::to_string(rc) + ": " + g_eng.last_error(ctx);
            }
            if (rc != BBP_OK) {
                logf(0, "Error resolving the request: %s", why.c_str());
                g_errors++;
                break;  // prove error -> no payload (main.rs:86-92 via try_result_future!), connection dropped
            }
            const tlv::Bytes out = frame(encode_proof(record.data(), proof_len, pr.n_items));
            if (!write_all(fd, out.data(), out.size())) break;
            logf(4, "Request resolved");
        } else if (opcode == OP_VERIFY) {
            VerifyRequest vr;
            uint8_t ok = 0;
            if (parse_verify_request(req.data() + 1, req.size() - 1, &vr, &why)) {
                const int32_t rc = g_eng.verify(ctx, vr.record.data(), (uint32_t)vr.record.size(), vr.score, vr.z_img, vr.seed,
                                                vr.pub_list.data(), vr.n_items);
                ok = rc == BBP_OK;
                if (rc > BBP_ERR_FORMAT) logf(1, "verify: engine status %d: %s", rc, g_eng.last_error(ctx));
            } else {
                logf(3, "verify request rejected while parsing: %s", why.c_str());
            }
            tlv::Bytes one(1, ok ? 0x01 : 0x00);  // main.rs:95-99: parse errors and verification failures both answer 0x00
            const tlv::Bytes out = frame(one);
            if (!write_all(fd, out.data(), out.size())) break;
            logf(4, "Request resolved");
        } else {
            logf(0, "Error resolving the request: Undefined operation code");
            g_errors++;
            break;
        }
        g_served++;
    }
    close(fd);
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live--;
    }
    g_live_cv.notify_all();
}

static void on_signal(int) {
    g_stop = 1;
    if (g_listen_fd >= 0) shutdown(g_listen_fd, SHUT_RDWR);  // wakes accept()
}

static void usage(const char* argv0) {
    fprintf(stderr,
            "usage: %s [-b|--bind-path PATH] [-l|--log-level error|warn|info|debug|trace] [--engine LIB.so] [--device N | --devices 0,1,..]\n"
            "          [--window-us US] [--max-batch B] [--max-connections C]\n",
            argv0);
}

int main(int argc, char** argv) {
    const char* tmp = getenv("TMPDIR");
    std::string bind_path = std::string(tmp && *tmp ? tmp : "/tmp") + "/dusk-uds-blindbid";  // src/main.rs:14-16
    std::string level = "info", engine_path;
    std::vector<int> devices;
    uint32_t window_us = 200, max_batch = 4096;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) {
                usage(argv[0]);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-b" || a == "--bind-path") bind_path = val();
        else if (a == "-l" || a == "--log-level") level = val();
        else if (a == "--engine") engine_path = val();
        else if (a == "--device") devices.assign(1, atoi(val()));
        else if (a == "--devices") {
            devices.clear();
            for (const char* p = val(); *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
        else if (a == "--window-us") window_us = (uint32_t)atoi(val());
        else if (a == "--max-batch") max_batch = (uint32_t)atoi(val());
        else if (a == "--max-connections") g_max_conn = atoi(val());
        else {
            usage(argv[0]);
            return 2;
        }
    }
    if (const char* rl = getenv("RUST_LOG")) level = rl;  // src/main.rs:45-47: the flag only fills RUST_LOG when it is unset
    if ((g_level = level_of(level.c_str())) < 0) {
        fprintf(stderr, "invalid log level '%s'\n", level.c_str());
        return 2;
    }
    if (engine_path.empty()) {  // libbbp_hip.so one directory up from this binary (dusk_blindbidproof_amd/)
        char self[4096];
        ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
        std::string dir = n > 0 ? std::string(self, (size_t)n) : std::string(argv[0]);
        dir = dir.substr(0, dir.find_last_of('/'));
        engine_path = dir.substr(0, dir.find_last_of('/')) + "/libbbp_hip.so";
    }
    std::string why;
    if (!g_eng.load(engine_path.c_str(), &why)) {
        logf(0, "cannot load engine %s: %s", engine_path.c_str(), why.c_str());
        return 1;
    }
    if (devices.empty()) devices.push_back(0);
    for (int device : devices) {
        bbp_ctx* c = nullptr;
        const int32_t rc = g_eng.init(device, &c);
        if (rc != BBP_OK) {
            logf(0, "bbp_init(device %d) failed with status %d: %s (this server has no CPU path)", device, rc, c ? g_eng.last_error(c) : "no usable device");
            return 1;
        }
        g_eng.set_batching(c, window_us, max_batch);
        g_eng.ctxs.push_back(c);
    }

    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_signal;
    sigaction(SIGTERM, &sa, nullptr);
    sigaction(SIGINT, &sa, nullptr);
    signal(SIGPIPE, SIG_IGN);

    g_listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (bind_path.size() >= sizeof addr.sun_path) {
        logf(0, "bind path too long");
        return 1;
    }
    strcpy(addr.sun_path, bind_path.c_str());
    unlink(bind_path.c_str());
    if (g_listen_fd < 0 || bind(g_listen_fd, (sockaddr*)&addr, sizeof addr) != 0 || listen(g_listen_fd, 1024) != 0) {
        logf(0, "Failed binding socket %s: %s", bind_path.c_str(), strerror(errno));
        return 1;
    }
    logf(2, "listening on %s (engine %s, %zu device context(s), batching window %u us, max batch %u)", bind_path.c_str(), engine_path.c_str(),
         g_eng.ctxs.size(), window_us, max_batch);
    while (!g_stop) {
        int fd = accept(g_listen_fd, nullptr, nullptr);
        if (fd < 0) {
            if (errno == EINTR) continue;
            break;
        }
        {
            std::unique_lock<std::mutex> lk(g_live_mu);
            g_live_cv.wait(lk, [] { return g_live < g_max_conn; });
            g_live++;
        }
        std::thread(serve, fd).detach();
    }
    {
        std::unique_lock<std::mutex> lk(g_live_mu);
        g_live_cv.wait_for(lk, std::chrono::seconds(5), [] { return g_live == 0; });
    }
    uint64_t calls = 0, reqs = 0;
    uint32_t biggest = 0;
    for (bbp_ctx* c : g_eng.ctxs) {
        uint64_t a = 0, b = 0;
        uint32_t m = 0;
        g_eng.batching_stats(c, &a, &b, &m);
        calls += a;
        reqs += b;
        if (m > biggest) biggest = m;
    }
    logf(2, "served %llu requests (%llu errors) in %llu device calls, largest batch %u", (unsigned long long)g_served.load(),
         (unsigned long long)g_errors.load(), (unsigned long long)calls, biggest);
    close(g_listen_fd);
    unlink(bind_path.c_str());
    if (g_live == 0)
        for (bbp_ctx* c : g_eng.ctxs) g_eng.free_(c);
    return 0;
}
