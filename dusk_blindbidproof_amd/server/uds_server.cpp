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
// What is different by design.  The reference runs each request on its own worker thread: N requests = N CPU proofs side by
// side.  Here NO thread belongs to a connection: a few I/O threads (--io-threads) each run an epoll loop over their share of the
// connections, assemble frames, parse them and hand the request to the engine with bbp_prove_async / bbp_verify_async; the
// engine's call combiner turns whatever is in flight into device batches (--window-us: how long a batch waits for company) and
// calls back when a request's batch is done; the callback only queues the result for the connection's I/O thread, which writes
// the reply.  (Round 2 had a thread per connection: 4096 connections = 4096 threads, and the socket path reached 76 % of the
// engine's rate.)  With --devices a,b,.. the engine handle is a device pool (include/bbp.h): one combiner deals the batches to
// the least-loaded GPU.  The engine is loaded with dlopen (--engine), so this binary holds no GPU code and the CPU-tier tests
// run it against a stub.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <strings.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/resource.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
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
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "[%s bbp-uds-server] %s\n", names[lvl], buf);
}

// ---- engine (the C ABI of include/bbp.h, bound at run time) ----------------------------------------------------------------------
struct Engine {
    void* so = nullptr;
    bbp_ctx* ctx = nullptr;  // one context, or a device pool over --devices (bids are independent: no cross-GPU traffic on this path)
    size_t n_devices = 0;
    decltype(&bbp_init) init = nullptr;
    decltype(&bbp_pool_init) pool_init = nullptr;
    decltype(&bbp_pool_member_stats) member_stats = nullptr;
    decltype(&bbp_free) free_ = nullptr;
    decltype(&bbp_prove_async) prove_async = nullptr;
    decltype(&bbp_verify_async) verify_async = nullptr;
    decltype(&bbp_last_error) last_error = nullptr;
    decltype(&bbp_proof_record_size) record_size = nullptr;
    decltype(&bbp_set_batching) set_batching = nullptr;
    decltype(&bbp_batching_stats) batching_stats = nullptr;
    decltype(&bbp_check_health) check_health = nullptr;
    decltype(&bbp_reserve) reserve = nullptr;
    decltype(&bbp_describe) describe = nullptr;  // optional
    bool load(const char* path, std::string* why) {
        so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!so) return *why = dlerror(), false;
        auto sym = [&](const char* n) {
            void* p = dlsym(so, n);
            if (!p && why->empty()) *why = std::string("missing symbol ") + n;
            return p;
        };
        init = (decltype(init))sym("bbp_init");
        pool_init = (decltype(pool_init))sym("bbp_pool_init");
        member_stats = (decltype(member_stats))sym("bbp_pool_member_stats");
        free_ = (decltype(free_))sym("bbp_free");
        prove_async = (decltype(prove_async))sym("bbp_prove_async");
        verify_async = (decltype(verify_async))sym("bbp_verify_async");
        last_error = (decltype(last_error))sym("bbp_last_error");
        record_size = (decltype(record_size))sym("bbp_proof_record_size");
        set_batching = (decltype(set_batching))sym("bbp_set_batching");
        batching_stats = (decltype(batching_stats))sym("bbp_batching_stats");
        check_health = (decltype(check_health))sym("bbp_check_health");
        reserve = (decltype(reserve))sym("bbp_reserve");
        describe = (decltype(describe))dlsym(so, "bbp_describe");
        return why->empty();
    }
};
static Engine g_eng;
static std::atomic<uint64_t> g_served{0}, g_errors{0}, g_next_id{1};
static std::atomic<int> g_live{0}, g_inflight{0};
static int g_max_conn = 65536;
static volatile sig_atomic_t g_stop = 0;
static std::atomic<int> g_fatal{0};  // the engine handle reported BBP_ERR_DEVICE: stop accepting, drain, exit non-zero
static int g_listen_fd = -1;
constexpr int EXIT_ENGINE_DEAD = 3;

constexpr uint64_t MAX_FRAME = 1u << 16;  // the largest legitimate request (verify, N = 202) is ~ 16 KB; per-connection buffering is bounded by 2 x this

// ---- one connection ---------------------------------------------------------------------------------------------------------------
struct Conn {
    int fd = -1;
    uint64_t id = 0;
    tlv::Bytes in;      // bytes received and not yet consumed
    tlv::Bytes out;     // reply bytes not yet written
    size_t out_off = 0;
    bool busy = false;  // a request of this connection is with the engine: replies keep request order, the next frame waits
    bool eof = false;   // the peer has closed its sending side: close once the pending reply is out
    bool want_out = false;
};

struct Reactor;
// a request on its way through the engine; owned by the engine callback, then by the connection's reactor
struct Pending {
    Reactor* reactor = nullptr;
    uint64_t conn_id = 0;
    uint8_t opcode = 0;
    uint32_t n_items = 0;
    tlv::Bytes record;  // prove: the engine writes the proof record here
    int32_t status = BBP_ERR_INTERNAL;
    std::string err;
};

struct Reactor {
    int ep = -1, evfd = -1;
    bool listening = false;
    std::thread th;
    std::mutex mu;
    std::vector<Pending*> done;  // engine callbacks -> this thread
    std::unordered_map<uint64_t, std::unique_ptr<Conn>> conns;

    void run();
    void accept_some();
    void on_readable(Conn* c);
    void process(Conn* c);
    bool dispatch(Conn* c, const uint8_t* req, size_t len);
    void complete(Pending* p);
    void flush(Conn* c);
    void drop(Conn* c);
    void set_listening(bool on);
};
static std::vector<std::unique_ptr<Reactor>> g_reactors;


// engine thread: hand the finished request to its connection's I/O thread and return at once (include/bbp.h bbp_done_fn)
static void on_engine_done(void* user, int32_t status) {
    Pending* p = static_cast<Pending*>(user);
    p->status = status;
    if (status != BBP_OK && status != BBP_ERR_VERIFY) p->err = g_eng.last_error(g_eng.ctx);  // this thread's slot holds the request's message now
    Reactor* r = p->reactor;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->done.push_back(p);
    }
    const uint64_t one = 1;
    (void)!write(r->evfd, &one, 8);
}

static const bool g_trace = getenv("BBP_TRACE") != nullptr;
static thread_local double t_submit_max_ms = 0, t_acc_ms = 0, t_comp_ms = 0, t_read_ms = 0;
static thread_local int t_submits = 0, t_events = 0;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void Reactor::set_listening(bool on) {
    if (on == listening || g_listen_fd < 0) return;
    epoll_event ev;
    memset(&ev, 0, sizeof ev);
    ev.events = EPOLLIN | EPOLLEXCLUSIVE;  // one I/O thread is woken per arriving connection
    ev.data.u64 = 0;
    if (epoll_ctl(ep, on ? EPOLL_CTL_ADD : EPOLL_CTL_DEL, g_listen_fd, &ev) == 0) listening = on;
}

void Reactor::drop(Conn* c) {
    epoll_ctl(ep, EPOLL_CTL_DEL, c->fd, nullptr);
    close(c->fd);
    conns.erase(c->id);  // (a request of this connection still with the engine finds no connection when it completes, and is dropped)
    g_live--;
}

void Reactor::accept_some() {
    for (int k = 0; k < 64; k++) {
        if (g_live.load() >= g_max_conn) {  // at the cap: arrivals wait in the listen backlog until a connection goes away
            set_listening(false);
            return;
        }
        const int fd = accept4(g_listen_fd, nullptr, nullptr, SOCK_NONBLOCK | SOCK_CLOEXEC);
        if (fd < 0) return;  // EAGAIN: another I/O thread took it, or nothing left
        auto c = std::make_unique<Conn>();
        c->fd = fd;
        c->id = g_next_id.fetch_add(1);
        epoll_event ev;
        memset(&ev, 0, sizeof ev);
        ev.events = EPOLLIN | EPOLLRDHUP;
        ev.data.u64 = c->id;
        if (epoll_ctl(ep, EPOLL_CTL_ADD, fd, &ev) != 0) {
            close(fd);
            continue;
        }
        g_live++;
        conns.emplace(c->id, std::move(c));
    }
}

void Reactor::flush(Conn* c) {
    while (c->out_off < c->out.size()) {
        const ssize_t w = send(c->fd, c->out.data() + c->out_off, c->out.size() - c->out_off, MSG_NOSIGNAL);
        if (w > 0) {
            c->out_off += (size_t)w;
            continue;
        }
        if (w < 0 && errno == EINTR) continue;
        if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
            if (!c->want_out) {
                epoll_event ev;
                memset(&ev, 0, sizeof ev);
                ev.events = EPOLLIN | EPOLLRDHUP | EPOLLOUT;
                ev.data.u64 = c->id;
                epoll_ctl(ep, EPOLL_CTL_MOD, c->fd, &ev);
                c->want_out = true;
            }
            return;
        }
        drop(c);  // the peer is gone
        return;
    }
    c->out.clear();
    c->out_off = 0;
    if (c->want_out) {
        epoll_event ev;
        memset(&ev, 0, sizeof ev);
        ev.events = EPOLLIN | EPOLLRDHUP;
        ev.data.u64 = c->id;
        epoll_ctl(ep, EPOLL_CTL_MOD, c->fd, &ev);
        c->want_out = false;
    }
}

// one request frame (payload of the outer TLV element, main.rs:70-79).  false = the connection is to be dropped without a reply
bool Reactor::dispatch(Conn* c, const uint8_t* req, size_t len) {
    if (len == 0) {
        logf(0, "Error resolving the request: The request was not provided");
        g_errors++;
        return false;
    }
    const uint8_t opcode = req[0];
    std::string why;
    if (opcode == OP_PROVE) {
        ProveRequest pr;
        if (!parse_prove_request(req + 1, len - 1, &pr, &why)) {
            logf(0, "Error resolving the request: %s", why.c_str());
            g_errors++;
            return false;  // prove error -> no payload (main.rs:86-92 via try_result_future!), connection dropped
        }
        Pending* p = new Pending();
        p->reactor = this;
        p->conn_id = c->id;
        p->opcode = opcode;
        p->n_items = pr.n_items;
        p->record.resize(g_eng.record_size(pr.n_items));
        g_inflight++;
        const double t0 = g_trace ? now_ms() : 0;
        const int32_t rc = g_eng.prove_async(g_eng.ctx, pr.scalars7, pr.pub_list.data(), pr.n_items, pr.toggle, nullptr, p->record.data(), on_engine_done, p);
        if (g_trace) {
            const double dt = now_ms() - t0;
            if (dt > t_submit_max_ms) t_submit_max_ms = dt;
            t_submits++;
        }
        if (rc != BBP_OK) {  // decided at once: the callback will not fire
            g_inflight--;
            logf(0, "Error resolving the request: engine status %d: %s", rc, g_eng.last_error(g_eng.ctx));
            delete p;
            g_errors++;
            return false;
        }
        c->busy = true;
        return true;
    }
    if (opcode == OP_VERIFY) {
        VerifyRequest vr;
        int32_t rc = BBP_ERR_FORMAT;
        if (parse_verify_request(req + 1, len - 1, &vr, &why)) {
            Pending* p = new Pending();
            p->reactor = this;
            p->conn_id = c->id;
            p->opcode = opcode;
            g_inflight++;
            rc = g_eng.verify_async(g_eng.ctx, vr.record.data(), (uint32_t)vr.record.size(), vr.score, vr.z_img, vr.seed, vr.pub_list.data(), vr.n_items,
                                    on_engine_done, p);
            if (rc == BBP_OK) {
                c->busy = true;
                return true;
            }
            g_inflight--;
            delete p;  // the structural parse on the host already decided: answered right here
            if (rc > BBP_ERR_FORMAT) logf(1, "verify: engine status %d: %s", rc, g_eng.last_error(g_eng.ctx));
        } else {
            logf(3, "verify request rejected while parsing: %s", why.c_str());
        }
        const tlv::Bytes reply = frame(tlv::Bytes(1, 0x00));  // main.rs:95-99: parse errors and verification failures both answer 0x00
        c->out.insert(c->out.end(), reply.begin(), reply.end());
        g_served++;
        logf(4, "Request resolved");
        return true;
    }
    logf(0, "Error resolving the request: Undefined operation code");
    g_errors++;
    return false;
}

// frames that are complete in c->in, in order, one engine request at a time
void Reactor::process(Conn* c) {
    size_t pos = 0;
    bool alive = true;
    while (alive && !c->busy) {
        const uint8_t* p = c->in.data() + pos;
        const size_t avail = c->in.size() - pos;
        if (!tlv::header_complete(p, avail)) break;
        uint64_t len;
        const size_t h = tlv::parse_header(p, avail, &len);
        if (!h || len > MAX_FRAME) {
            logf(0, "Error resolving the request: unreadable frame");
            g_errors++;
            alive = false;
            break;
        }
        if (avail - h < len) break;
        alive = dispatch(c, p + h, (size_t)len);
        pos += h + (size_t)len;
    }
    if (!alive) {
        drop(c);
        return;
    }
    if (pos) c->in.erase(c->in.begin(), c->in.begin() + (ptrdiff_t)pos);
    if (!c->out.empty()) flush(c);
}

void Reactor::on_readable(Conn* c) {
    uint8_t buf[16384];
    for (;;) {
        const ssize_t r = read(c->fd, buf, sizeof buf);
        if (r > 0) {
            if (c->in.size() + (size_t)r > 2 * MAX_FRAME + 16) {  // a peer that keeps sending while its request is with the engine
                drop(c);
                return;
            }
            c->in.insert(c->in.end(), buf, buf + r);
            if ((size_t)r < sizeof buf) break;
            continue;
        }
        if (r < 0 && errno == EINTR) continue;
        if (r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) break;
        c->eof = true;  // 0 = the peer closed (or an error: same treatment)
        break;
    }
    const uint64_t id = c->id;
    process(c);
    auto it = conns.find(id);
    if (it == conns.end()) return;  // dropped while processing
    c = it->second.get();
    if (c->eof && !c->busy && c->out.empty()) {
        if (!c->in.empty()) {
            logf(0, "Error resolving the request: unreadable frame");  // the stream ended inside a frame
            g_errors++;
        }
        drop(c);
    }
}

// a request is back from the engine: write the reply (this thread owns the connection), then look at the next frame
void Reactor::complete(Pending* p) {
    std::unique_ptr<Pending> own(p);
    g_inflight--;
    auto it = conns.find(p->conn_id);
    if (p->status == BBP_ERR_DEVICE) {
        // The engine's health word is sticky: from here on EVERY call on this handle comes back BBP_ERR_DEVICE ("free this context and
        // create a new one", include/bbp.h).  The request was NOT judged, so a verify gets no 0x00 (that byte means "rejected") and a
        // prove gets nothing, as for any prove error: the connection is dropped.  The process stops accepting, drains what is in
        // flight and exits with EXIT_ENGINE_DEAD so that its supervisor starts a fresh one (a process that has initialised the GPU
        // is never re-exec'ed; re-initialising the pool in place would keep a possibly damaged device mapping).
        if (!g_fatal.exchange(1)) {
            logf(0, "engine reports a device failure: %s -- no further requests are accepted; exiting with status %d once in-flight requests have drained",
                 p->err.c_str(), EXIT_ENGINE_DEAD);
            g_stop = 1;
            if (g_listen_fd >= 0) shutdown(g_listen_fd, SHUT_RDWR);
        }
        g_errors++;
        if (it != conns.end()) drop(it->second.get());
        return;
    }
    if (it == conns.end()) return;  // the peer went away meanwhile
    Conn* c = it->second.get();
    c->busy = false;
    if (p->opcode == OP_PROVE) {
        if (p->status != BBP_OK) {
            logf(0, "Error resolving the request: engine status %d: %s", p->status, p->err.c_str());
            g_errors++;
            drop(c);  // prove error -> nothing written
            return;
        }
        const tlv::Bytes reply = frame(encode_proof(p->record.data(), BBP_R1CS_PROOF_BYTES, p->n_items));
        c->out.insert(c->out.end(), reply.begin(), reply.end());
    } else {
        if (p->status > BBP_ERR_FORMAT) logf(1, "verify: engine status %d: %s", p->status, p->err.c_str());
        const tlv::Bytes reply = frame(tlv::Bytes(1, p->status == BBP_OK ? 0x01 : 0x00));
        c->out.insert(c->out.end(), reply.begin(), reply.end());
    }
    g_served++;
    logf(4, "Request resolved");
    const uint64_t id = c->id;
    flush(c);
    if (conns.find(id) == conns.end()) return;
    process(c);
    auto again = conns.find(id);
    if (again == conns.end()) return;
    c = again->second.get();
    if (c->eof && !c->busy && c->out.empty()) drop(c);
}

void Reactor::run() {
    std::vector<epoll_event> evs(512);
    std::vector<Pending*> batch;
    double t_iter = now_ms();
    while (!g_stop) {
        if (g_trace) {  // BBP_TRACE: an iteration of this loop that took long is time during which this thread's connections were not served
            const double t = now_ms();
            if (t - t_iter > 10.0)
                fprintf(stderr, "[bbp trace] reactor %p: iteration took %.1f ms (%d events: accept %.1f ms, completions %.1f ms, reads %.1f ms; %d engine submits, slowest %.1f ms) at %.1f\n",
                        (void*)this, t - t_iter, t_events, t_acc_ms, t_comp_ms, t_read_ms, t_submits, t_submit_max_ms, t);
            t_submit_max_ms = t_acc_ms = t_comp_ms = t_read_ms = 0;
            t_submits = t_events = 0;
        }
        const int n = epoll_wait(ep, evs.data(), (int)evs.size(), 200);
        t_iter = now_ms();
        t_events = n;
        if (n < 0 && errno != EINTR) break;
        for (int i = 0; i < n; i++) {
            const uint64_t id = evs[i].data.u64;
            if (id == 0) {
                const double a0 = g_trace ? now_ms() : 0;
                accept_some();
                if (g_trace) t_acc_ms += now_ms() - a0;
                continue;
            }
            if (id == UINT64_MAX) {  // completions from the engine
                uint64_t cnt;
                (void)!read(evfd, &cnt, 8);
                batch.clear();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    batch.swap(done);
                }
                const double c0 = g_trace ? now_ms() : 0;
                for (Pending* p : batch) complete(p);
                if (g_trace) t_comp_ms += now_ms() - c0;
                continue;
            }
            auto it = conns.find(id);
            if (it == conns.end()) continue;
            Conn* c = it->second.get();
            if (evs[i].events & EPOLLOUT) {
                flush(c);
                if (conns.find(id) == conns.end()) continue;
                if (c->eof && !c->busy && c->out.empty()) {
                    drop(c);
                    continue;
                }
            }
            if (evs[i].events & (EPOLLIN | EPOLLRDHUP | EPOLLHUP | EPOLLERR)) {
                if ((evs[i].events & (EPOLLHUP | EPOLLERR)) && !(evs[i].events & EPOLLIN)) {
                    c->eof = true;
                    if (!c->busy) drop(c);
                    else epoll_ctl(ep, EPOLL_CTL_DEL, c->fd, nullptr);  // the pending completion finds it and drops it
                    continue;
                }
                if (c->eof) {  // already seen: nothing more to read, the level-triggered RDHUP would spin
                    epoll_event ev;
                    memset(&ev, 0, sizeof ev);
                    ev.events = c->want_out ? EPOLLOUT : 0;
                    ev.data.u64 = c->id;
                    epoll_ctl(ep, EPOLL_CTL_MOD, c->fd, &ev);
                    continue;
                }
                const double r0 = g_trace ? now_ms() : 0;
                on_readable(c);
                if (g_trace) t_read_ms += now_ms() - r0;
            }
        }
        if (!listening && !g_stop && g_live.load() < g_max_conn) set_listening(true);
    }
    // stopping: no new requests; replies of requests still with the engine are dropped with their connections
    std::vector<uint64_t> ids;
    for (auto& kv : conns) ids.push_back(kv.first);
    for (uint64_t id : ids) {
        auto it = conns.find(id);
        if (it != conns.end()) drop(it->second.get());
    }
}

static void on_signal(int) {
    g_stop = 1;
    if (g_listen_fd >= 0) shutdown(g_listen_fd, SHUT_RDWR);
}

static void usage(const char* argv0) {
    fprintf(stderr,
            "usage: %s [-b|--bind-path PATH] [-l|--log-level error|warn|info|debug|trace] [--engine LIB.so] [--device N | --devices 0,1,.. | --devices all]\n"
            "          [--window-us US] [--max-batch B] [--max-connections C] [--io-threads T] [--reserve N[,N..]] [--verify-aggregate G]\n",
            argv0);
}

int main(int argc, char** argv) {
    const char* tmp = getenv("TMPDIR");
    std::string bind_path = std::string(tmp && *tmp ? tmp : "/tmp") + "/dusk-uds-blindbid";  // src/main.rs:14-16
    std::string level = "info", engine_path;
    std::vector<int32_t> devices, reserve_items;  // --reserve: bid-list lengths whose buffers are sized for --max-batch before the first request
    uint32_t window_us = 200, max_batch = 4096;
    int io_threads = 2;
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
            const char* list = val();
            if (strcmp(list, "all") == 0) {  // every visible GPU: bbp_init(-1) = bbp_init_all
                devices.assign(1, -1);
                continue;
            }
            for (const char* p = list; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
        else if (a == "--window-us") window_us = (uint32_t)atoi(val());
        else if (a == "--max-batch") max_batch = (uint32_t)atoi(val());
        else if (a == "--max-connections") g_max_conn = atoi(val());
        else if (a == "--io-threads") io_threads = atoi(val());
        else if (a == "--verify-aggregate") setenv("BBP_VERIFY_AGGREGATE", val(), 1);  // opcode-2 batches of >= 2 G proofs are checked in groups of G with
                                                                                         // per-proof fallback: same verdicts, 2-3x the rate (include/bbp.h)
        else if (a == "--reserve") {
            for (const char* p = val(); *p;) {
                reserve_items.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
        else {
            usage(argv[0]);
            return 2;
        }
    }
    if (io_threads < 1) io_threads = 1;
    if (io_threads > 64) io_threads = 64;
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
    // The engine creates eleven streams per context and keeps up to eight busy: HIP's default of four hardware queues makes them share
    // and serialise.  Eight is the measured optimum for this server's mix (prove-dominated, host-pointer calls: 20.4 k proofs/s at 3072
    // connections against 17.1 k with 16 queues; prove + verify at 2048 connections 12.4-12.9 k ops/s against 12.4-12.5 k) -- the
    // verifier's four lanes alone would like 16.  Read when the HIP runtime initialises, i.e. at the engine's
    // first call: set before the library is loaded, and only if the operator has not chosen.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    std::string why;
    if (!g_eng.load(engine_path.c_str(), &why)) {
        logf(0, "cannot load engine %s: %s", engine_path.c_str(), why.c_str());
        return 1;
    }
    if (devices.empty()) devices.push_back(0);
    {
        const int32_t rc = devices.size() == 1 ? g_eng.init(devices[0], &g_eng.ctx) : g_eng.pool_init(devices.data(), (uint32_t)devices.size(), &g_eng.ctx);
        if (rc != BBP_OK) {
            logf(0, "engine initialisation on %zu device(s) failed with status %d: %s (this server has no CPU path)", devices.size(), rc,
                 g_eng.ctx ? g_eng.last_error(g_eng.ctx) : "no usable device");
            return 1;
        }
        g_eng.n_devices = devices.size();
        if (devices.size() == 1 && devices[0] == -1) {  // --devices all: ask the pool how many it found
            auto pool_size = (decltype(&bbp_pool_size))dlsym(g_eng.so, "bbp_pool_size");
            g_eng.n_devices = pool_size ? pool_size(g_eng.ctx) : 1;
            devices.resize(g_eng.n_devices);
            for (size_t i = 0; i < devices.size(); i++) devices[i] = (int32_t)i;
        }
        g_eng.set_batching(g_eng.ctx, window_us, max_batch);
        if (g_eng.describe) {  // what the engine runs on; its WARNING lines (hardware queues, memory) at warn level
            static char report[8192];
            if (g_eng.describe(g_eng.ctx, report, sizeof report) == BBP_OK)
                for (char* ln = strtok(report, "\n"); ln; ln = strtok(nullptr, "\n")) logf(strncmp(ln, "WARNING", 7) == 0 ? 1 : 2, "engine: %s", ln);
        }
        for (int32_t n : reserve_items) {
            const int32_t rr = g_eng.reserve(g_eng.ctx, max_batch, (uint32_t)n);
            if (rr != BBP_OK) logf(1, "--reserve %d: engine status %d: %s (buffers will grow on demand)", n, rr, g_eng.last_error(g_eng.ctx));
            else logf(2, "buffers sized for batches of %u with bid lists of %d", max_batch, n);
        }
    }
    {  // every connection is a descriptor: lift the soft limit to the hard one
        rlimit rl;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < rl.rlim_max) {
            rl.rlim_cur = rl.rlim_max;
            setrlimit(RLIMIT_NOFILE, &rl);
        }
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur != RLIM_INFINITY && (rlim_t)g_max_conn + 64 > rl.rlim_cur) g_max_conn = (int)rl.rlim_cur - 64;
        // Grow the descriptor table to its final size NOW.  The kernel doubles it on demand, and in a multi-threaded process every
        // doubling waits for an RCU grace period inside accept4 (expand_fdtable -> synchronize_rcu): measured here as nine stalls of
        // 120-190 ms each -- both I/O threads frozen in accept, no request read -- while the first 16 k connections arrived.
        const int hi = fcntl(0, F_DUPFD_CLOEXEC, g_max_conn + 32);
        if (hi >= 0) close(hi);
    }

    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_signal;
    sigaction(SIGTERM, &sa, nullptr);
    sigaction(SIGINT, &sa, nullptr);
    signal(SIGPIPE, SIG_IGN);

    g_listen_fd = socket(AF_UNIX, SOCK_STREAM | SOCK_NONBLOCK | SOCK_CLOEXEC, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (bind_path.size() >= sizeof addr.sun_path) {
        logf(0, "bind path too long");
        return 1;
    }
    strcpy(addr.sun_path, bind_path.c_str());
    unlink(bind_path.c_str());
    if (g_listen_fd < 0 || bind(g_listen_fd, (sockaddr*)&addr, sizeof addr) != 0 || listen(g_listen_fd, 4096) != 0) {
        logf(0, "Failed binding socket %s: %s", bind_path.c_str(), strerror(errno));
        return 1;
    }
    for (int t = 0; t < io_threads; t++) {
        auto r = std::make_unique<Reactor>();
        r->ep = epoll_create1(EPOLL_CLOEXEC);
        r->evfd = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
        epoll_event ev;
        memset(&ev, 0, sizeof ev);
        ev.events = EPOLLIN;
        ev.data.u64 = UINT64_MAX;
        if (r->ep < 0 || r->evfd < 0 || epoll_ctl(r->ep, EPOLL_CTL_ADD, r->evfd, &ev) != 0) {
            logf(0, "cannot set up an I/O thread: %s", strerror(errno));
            return 1;
        }
        r->set_listening(true);
        g_reactors.push_back(std::move(r));
    }
    logf(2, "listening on %s (engine %s, %zu device context(s), batching window %u us, max batch %u, %d I/O thread(s))", bind_path.c_str(),
         engine_path.c_str(), g_eng.n_devices, window_us, max_batch, io_threads);
    try {
        for (auto& r : g_reactors) r->th = std::thread([p = r.get()] { p->run(); });
    } catch (const std::exception& e) {
        logf(0, "cannot start an I/O thread: %s", e.what());
        g_stop = 1;
    }
    for (auto& r : g_reactors)
        if (r->th.joinable()) r->th.join();

    // every connection is closed now; requests still with the engine finish on their own (their replies have nowhere to go).
    // Wait for them -- bbp_free must not run under a batch -- but not for ever.
    for (int i = 0; i < 1000; i++) {
        for (auto& r : g_reactors) {  // completions that arrive now that the I/O threads have left
            std::vector<Pending*> late;
            {
                std::lock_guard<std::mutex> lk(r->mu);
                late.swap(r->done);
            }
            for (Pending* p : late) {
                delete p;
                g_inflight--;
            }
        }
        if (g_inflight.load() <= 0) break;
        usleep(10000);
    }
    uint64_t calls = 0, reqs = 0;
    uint32_t biggest = 0;
    g_eng.batching_stats(g_eng.ctx, &calls, &reqs, &biggest);
    logf(2, "served %llu requests (%llu errors) in %llu device calls, largest batch %u", (unsigned long long)g_served.load(),
         (unsigned long long)g_errors.load(), (unsigned long long)calls, biggest);
    if (g_eng.n_devices > 1)
        for (size_t i = 0; i < g_eng.n_devices; i++) {
            uint64_t c = 0, q = 0;
            g_eng.member_stats(g_eng.ctx, (uint32_t)i, &c, &q);
            logf(2, "device context %zu (device %d): %llu device calls, %llu requests", i, devices[i], (unsigned long long)c, (unsigned long long)q);
        }
    uint32_t health = 0;
    if (g_eng.check_health(g_eng.ctx, &health) == BBP_OK && health) logf(0, "engine health flags %#x at shutdown: results since the flag was raised may be wrong", health);
    close(g_listen_fd);
    unlink(bind_path.c_str());
    if (g_inflight.load() > 0) {
        logf(0, "%d request(s) still with the engine after 10 s: leaving without tearing the engine down", g_inflight.load());
        fflush(stderr);
        _exit(1);  // static destructors / HIP teardown under running GPU work can hang or abort
    }
    if (g_fatal.load()) {
        // the handle reported a device failure: tearing HIP state down on a device in an unknown condition can hang; the process is
        // about to be replaced anyway
        logf(0, "leaving with status %d without tearing the engine down (device failure)", EXIT_ENGINE_DEAD);
        fflush(stderr);
        _exit(EXIT_ENGINE_DEAD);
    }
    g_eng.free_(g_eng.ctx);
    return 0;
}
