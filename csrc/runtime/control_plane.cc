// TCP control plane of the cluster runtime.
//
// The reference's tasks talk through TensorFlow's in-process gRPC server
// (/root/reference/example.py:128-131: tf.train.Server + server.join()).  Bulk tensors here
// travel over RCCL/xGMI; what is left for TCP is small and latency-tolerant: rendezvous
// keys, request queues of the parameter-server tasks, the shared global_step, staleness
// clocks, barriers, heartbeats and stop requests.  One server lives in every "ps" task (at
// its host:port from the cluster spec) and in the chief for ps-less clusters.
//
// Wire format (little endian): request  = u32 body_len | u8 op | body
//                              response = u32 body_len | i32 status | body
// Strings/blobs inside a body are u32 length + bytes.  One thread per connection.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "crc32c.h"

namespace {

enum Op : uint8_t {
    OP_SET = 1, OP_GET = 2, OP_ADD = 3, OP_PUSH = 4, OP_POP = 5, OP_CLOCK_SET = 6, OP_CLOCK_WAIT = 7,
    OP_BARRIER = 8, OP_HEARTBEAT = 9, OP_DEAD = 10, OP_CLOCK_MIN = 11, OP_DELETE = 12, OP_QLEN = 13,
    // folded operations: what used to be several round trips of one training step is one each
    OP_TAKE_MANY = 14,        // block until every key exists, return all values, erase them
    OP_PUSH_MANY = 15,        // append one value to each of several queues
    OP_CLOCK_SET_WAIT = 16,   // publish my clock, then block until mine - min(group) <= bound
    OP_WAIT_COUNTER = 17,     // block until a counter has reached a value
    OP_ADD_SET = 18,          // add to a counter and set a key (the counter's new value is returned)
};
// A frame larger than this is a protocol error (or not one of our clients): the connection is dropped instead of
// allocating what its 4-byte length asks for.
constexpr uint32_t MAX_FRAME_BYTES = 64u << 20;
enum Status : int32_t { ST_OK = 0, ST_TIMEOUT = 1, ST_ERROR = -1, ST_CLOSED = -2 };

using Clock = std::chrono::steady_clock;

bool read_full(int fd, void* buf, size_t n) {
    auto* p = static_cast<uint8_t*>(buf);
    while (n) {
        ssize_t r = ::recv(fd, p, n, 0);
        if (r <= 0) {
            if (r < 0 && errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= static_cast<size_t>(r);
    }
    return true;
}

bool write_full(int fd, const void* buf, size_t n) {
    auto* p = static_cast<const uint8_t*>(buf);
    while (n) {
        ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
        if (r <= 0) {
            if (r < 0 && errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= static_cast<size_t>(r);
    }
    return true;
}

struct Writer {
    std::string b;
    void u8(uint8_t v) { b.push_back(static_cast<char>(v)); }
    void u32(uint32_t v) { b.append(reinterpret_cast<const char*>(&v), 4); }
    void i64(int64_t v) { b.append(reinterpret_cast<const char*>(&v), 8); }
    void str(const std::string& s) { u32(static_cast<uint32_t>(s.size())); b.append(s); }
    void blob(const void* p, size_t n) { u32(static_cast<uint32_t>(n)); b.append(static_cast<const char*>(p), n); }
};

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    Reader(const std::string& s) : p(reinterpret_cast<const uint8_t*>(s.data())), end(p + s.size()) {}
    bool need(size_t n) { if (static_cast<size_t>(end - p) < n) { ok = false; return false; } return true; }
    uint32_t u32() { uint32_t v = 0; if (need(4)) { std::memcpy(&v, p, 4); p += 4; } return v; }
    int64_t i64() { int64_t v = 0; if (need(8)) { std::memcpy(&v, p, 8); p += 8; } return v; }
    std::string str() { uint32_t n = u32(); std::string s; if (need(n)) { s.assign(reinterpret_cast<const char*>(p), n); p += n; } return s; }
};

// ---------------------------------------------------------------------------------- server state
struct State {
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::string, std::string> kv;
    std::map<std::string, int64_t> counters;
    std::map<std::string, std::deque<std::string>> queues;
    std::map<std::string, std::map<int64_t, int64_t>> clocks;     // group -> member -> value
    struct Barrier { int64_t arrived = 0; int64_t generation = 0; };
    std::map<std::string, Barrier> barriers;
    std::map<std::string, Clock::time_point> heartbeats;
    bool stopping = false;
};

int64_t clock_min(const std::map<int64_t, int64_t>& g) {
    int64_t m = INT64_MAX;
    for (auto& kv : g) m = kv.second < m ? kv.second : m;
    return g.empty() ? 0 : m;
}

template <class Pred>
bool wait_until(State& st, std::unique_lock<std::mutex>& lk, int64_t timeout_ms, Pred pred) {
    if (timeout_ms < 0) {
        st.cv.wait(lk, [&] { return st.stopping || pred(); });
        return pred();
    }
    // Deadline on the steady clock, sleeping in slices of pthread_cond_timedwait (system clock): a wall-clock jump
    // distorts at most one slice, and ThreadSanitizer sees the mutex hand-off (GCC 11's runtime does not intercept
    // the steady-clock wait that wait_for would use and then reports false races).
    const auto deadline = Clock::now() + std::chrono::milliseconds(timeout_ms);
    while (!(st.stopping || pred())) {
        const auto left = deadline - Clock::now();
        if (left <= Clock::duration::zero()) break;
        const auto slice = std::min<Clock::duration>(left, std::chrono::milliseconds(200));
        st.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::microseconds>(slice));
    }
    return pred();
}

struct Server {
    State st;
    int listen_fd = -1;
    int port = 0;
    std::thread accept_thread;
    std::mutex conn_mu;
    struct Conn { std::thread th; int fd = -1; bool done = false; };
    std::map<uint64_t, Conn> conns;            // live (and finished, not yet joined) connections by id
    uint64_t next_conn = 0;

    void handle(uint64_t id, int fd) {
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        try {
            for (;;) {
                uint32_t len;
                uint8_t op;
                if (!read_full(fd, &len, 4) || !read_full(fd, &op, 1)) break;
                if (len > MAX_FRAME_BYTES) break;                   // not a frame of ours: drop the connection
                std::string body(len, '\0');
                if (len && !read_full(fd, &body[0], len)) break;
                Reader r(body);
                Writer w;
                int32_t status = ST_OK;
                dispatch(op, r, w, status);
                if (!r.ok) status = ST_ERROR;
                uint32_t blen = static_cast<uint32_t>(w.b.size());
                if (!write_full(fd, &blen, 4) || !write_full(fd, &status, 4) || (blen && !write_full(fd, w.b.data(), blen))) break;
            }
        } catch (...) {
            // a failed allocation or a malformed body ends this connection only, never the task
        }
        // forget the descriptor BEFORE closing it: stop() must never shut down a number the process has reused
        std::lock_guard<std::mutex> g(conn_mu);
        auto it = conns.find(id);
        if (it != conns.end()) { it->second.fd = -1; it->second.done = true; }
        ::close(fd);
    }

    // joins the threads of connections that have ended (called with conn_mu held, from the accept thread)
    void reap_locked() {
        for (auto it = conns.begin(); it != conns.end();) {
            if (it->second.done) {
                if (it->second.th.joinable()) it->second.th.join();
                it = conns.erase(it);
            } else {
                ++it;
            }
        }
    }

    void dispatch(uint8_t op, Reader& r, Writer& w, int32_t& status) {
        std::unique_lock<std::mutex> lk(st.mu);
        switch (op) {
            case OP_SET: {
                std::string k = r.str(), v = r.str();
                st.kv[k] = v;
                st.cv.notify_all();
                break;
            }
            case OP_GET: {
                std::string k = r.str();
                int64_t to = r.i64();
                if (!wait_until(st, lk, to, [&] { return st.kv.count(k) > 0; })) { status = ST_TIMEOUT; break; }
                w.str(st.kv[k]);
                break;
            }
            case OP_DELETE: {
                std::string k = r.str();
                st.kv.erase(k);
                st.counters.erase(k);
                break;
            }
            case OP_ADD: {
                std::string k = r.str();
                int64_t d = r.i64();
                int64_t v = (st.counters[k] += d);
                w.i64(v);
                st.cv.notify_all();
                break;
            }
            case OP_PUSH: {
                std::string q = r.str(), v = r.str();
                st.queues[q].push_back(v);
                st.cv.notify_all();
                break;
            }
            case OP_POP: {
                std::string q = r.str();
                int64_t to = r.i64();
                if (!wait_until(st, lk, to, [&] { return !st.queues[q].empty(); })) { status = ST_TIMEOUT; break; }
                w.str(st.queues[q].front());
                st.queues[q].pop_front();
                break;
            }
            case OP_QLEN: {
                std::string q = r.str();
                w.i64(static_cast<int64_t>(st.queues[q].size()));
                break;
            }
            case OP_CLOCK_SET: {
                std::string g = r.str();
                int64_t member = r.i64(), value = r.i64();
                st.clocks[g][member] = value;
                w.i64(clock_min(st.clocks[g]));
                st.cv.notify_all();
                break;
            }
            case OP_CLOCK_MIN: {
                std::string g = r.str();
                w.i64(clock_min(st.clocks[g]));
                break;
            }
            case OP_CLOCK_WAIT: {
                // block until my_value - min(group) <= bound  (stale-synchronous-parallel gate)
                std::string g = r.str();
                int64_t mine = r.i64(), bound = r.i64(), to = r.i64();
                auto ok = [&] { return mine - clock_min(st.clocks[g]) <= bound; };
                if (!wait_until(st, lk, to, ok)) { status = ST_TIMEOUT; }
                w.i64(clock_min(st.clocks[g]));
                break;
            }
            case OP_TAKE_MANY: {
                uint32_t n = r.u32();
                std::vector<std::string> keys;
                for (uint32_t i = 0; i < n && r.ok; ++i) keys.push_back(r.str());
                int64_t to = r.i64();
                if (!r.ok) break;
                auto all = [&] { for (auto& k : keys) if (!st.kv.count(k)) return false; return true; };
                if (!wait_until(st, lk, to, all)) { status = ST_TIMEOUT; break; }
                w.u32(n);
                for (auto& k : keys) { w.str(st.kv[k]); st.kv.erase(k); }
                break;
            }
            case OP_PUSH_MANY: {
                uint32_t n = r.u32();
                for (uint32_t i = 0; i < n && r.ok; ++i) {
                    std::string q = r.str(), v = r.str();
                    if (r.ok) st.queues[q].push_back(v);
                }
                st.cv.notify_all();
                break;
            }
            case OP_CLOCK_SET_WAIT: {
                std::string g = r.str();
                int64_t member = r.i64(), value = r.i64(), bound = r.i64(), to = r.i64();
                st.clocks[g][member] = value;
                st.cv.notify_all();
                auto ok = [&] { return bound < 0 || value - clock_min(st.clocks[g]) <= bound; };
                if (!wait_until(st, lk, to, ok)) { status = ST_TIMEOUT; }
                w.i64(clock_min(st.clocks[g]));
                break;
            }
            case OP_WAIT_COUNTER: {
                std::string k = r.str();
                int64_t want = r.i64(), to = r.i64();
                if (!wait_until(st, lk, to, [&] { return st.counters[k] >= want; })) { status = ST_TIMEOUT; }
                w.i64(st.counters[k]);
                break;
            }
            case OP_ADD_SET: {
                std::string ck = r.str();
                int64_t d = r.i64();
                std::string k = r.str(), v = r.str();
                int64_t nv = ck.empty() ? 0 : (st.counters[ck] += d);
                // "%GS%" inside the value is replaced by the counter's new value (a reply that carries the step it made)
                auto pos = v.find("%GS%");
                if (pos != std::string::npos) v.replace(pos, 4, std::to_string(nv));
                st.kv[k] = v;
                w.i64(nv);
                st.cv.notify_all();
                break;
            }
            case OP_BARRIER: {
                std::string name = r.str();
                int64_t n = r.i64(), to = r.i64();
                auto& b = st.barriers[name];
                int64_t gen = b.generation;
                if (++b.arrived >= n) {
                    b.arrived = 0;
                    ++b.generation;
                    st.cv.notify_all();
                } else if (!wait_until(st, lk, to, [&] { return st.barriers[name].generation != gen; })) {
                    status = ST_TIMEOUT;
                    auto& bb = st.barriers[name];
                    if (bb.generation == gen && bb.arrived > 0) --bb.arrived;
                }
                break;
            }
            case OP_HEARTBEAT: {
                std::string who = r.str();
                st.heartbeats[who] = Clock::now();
                break;
            }
            case OP_DEAD: {
                // members whose last heartbeat is older than max_age_ms (newline separated)
                int64_t max_age = r.i64();
                std::string out;
                auto now = Clock::now();
                for (auto& kv : st.heartbeats) {
                    auto age = std::chrono::duration_cast<std::chrono::milliseconds>(now - kv.second).count();
                    if (age > max_age) { out += kv.first; out += '\n'; }
                }
                w.str(out);
                break;
            }
            default:
                status = ST_ERROR;
        }
    }

    bool start(const char* host, int want_port) {
        listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (listen_fd < 0) return false;
        int one = 1;
        setsockopt(listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in addr{};
        addr.sin_family = AF_INET;
        addr.sin_port = htons(static_cast<uint16_t>(want_port));
        addr.sin_addr.s_addr = htonl(INADDR_ANY);
        if (host && *host && std::strcmp(host, "0.0.0.0") != 0) {
            in_addr a{};
            if (inet_pton(AF_INET, host, &a) == 1) addr.sin_addr = a;
            else if (std::strcmp(host, "localhost") == 0) addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
        }
        if (::bind(listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0 || ::listen(listen_fd, 256) != 0) {
            ::close(listen_fd);
            listen_fd = -1;
            return false;
        }
        socklen_t sl = sizeof(addr);
        getsockname(listen_fd, reinterpret_cast<sockaddr*>(&addr), &sl);
        port = ntohs(addr.sin_port);
        const int lfd = listen_fd;       // the accept thread works on its own copy; stop() owns the member
        accept_thread = std::thread([this, lfd] {
            for (;;) {
                int fd = ::accept(lfd, nullptr, nullptr);
                if (fd < 0) {
                    if (errno == EINTR) continue;
                    break;
                }
                std::lock_guard<std::mutex> g(conn_mu);
                reap_locked();
                const uint64_t id = next_conn++;
                Conn& c = conns[id];
                c.fd = fd;
                c.th = std::thread([this, id, fd] { handle(id, fd); });
            }
        });
        return true;
    }

    void stop() {
        {
            std::lock_guard<std::mutex> g(st.mu);
            st.stopping = true;
            st.cv.notify_all();
        }
        if (listen_fd >= 0) ::shutdown(listen_fd, SHUT_RDWR);      // wakes accept(); the descriptor stays valid until
        if (accept_thread.joinable()) accept_thread.join();        // the accept thread is gone
        if (listen_fd >= 0) {
            ::close(listen_fd);
            listen_fd = -1;
        }
        std::vector<std::thread> threads;
        {
            std::lock_guard<std::mutex> g(conn_mu);
            for (auto& kv : conns) {
                if (kv.second.fd >= 0) ::shutdown(kv.second.fd, SHUT_RDWR);   // still open: owned by its thread
                threads.push_back(std::move(kv.second.th));
            }
        }
        for (auto& t : threads)
            if (t.joinable()) t.join();
        std::lock_guard<std::mutex> g(conn_mu);
        conns.clear();
    }
};

// ---------------------------------------------------------------------------------- client
struct Client {
    int fd = -1;
    std::mutex mu;
    std::string last;   // body of the last response

    int32_t call(uint8_t op, const Writer& w) {
        std::lock_guard<std::mutex> g(mu);
        if (fd < 0) return ST_CLOSED;
        uint32_t len = static_cast<uint32_t>(w.b.size());
        if (!write_full(fd, &len, 4) || !write_full(fd, &op, 1) || (len && !write_full(fd, w.b.data(), len))) return ST_CLOSED;
        uint32_t blen;
        int32_t status;
        if (!read_full(fd, &blen, 4) || !read_full(fd, &status, 4)) return ST_CLOSED;
        last.assign(blen, '\0');
        if (blen && !read_full(fd, &last[0], blen)) return ST_CLOSED;
        return status;
    }
};

int copy_blob(const std::string& body, void* buf, int64_t cap, int64_t* len) {
    Reader r(body);
    std::string s = r.str();
    if (!r.ok) return ST_ERROR;
    if (len) *len = static_cast<int64_t>(s.size());
    if (static_cast<int64_t>(s.size()) > cap) return ST_ERROR;
    if (!s.empty()) std::memcpy(buf, s.data(), s.size());
    return ST_OK;
}

}  // namespace

DTF_EXPORT void* dtf_cp_server_start(const char* host, int port) {
    auto* s = new Server();
    if (!s->start(host, port)) {
        delete s;
        return nullptr;
    }
    return s;
}
DTF_EXPORT int dtf_cp_server_port(void* h) { return h ? static_cast<Server*>(h)->port : -1; }
DTF_EXPORT void dtf_cp_server_stop(void* h) {
    if (!h) return;
    auto* s = static_cast<Server*>(h);
    s->stop();
    delete s;
}

DTF_EXPORT void* dtf_cp_connect(const char* host, int port, double timeout_s) {
    auto deadline = Clock::now() + std::chrono::milliseconds(static_cast<int64_t>(timeout_s * 1000));
    std::string h = (host && *host) ? host : "127.0.0.1";
    if (h == "localhost") h = "127.0.0.1";
    for (;;) {
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        std::string ps = std::to_string(port);
        if (getaddrinfo(h.c_str(), ps.c_str(), &hints, &res) == 0 && res) {
            int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
            if (fd >= 0) {
                if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
                    freeaddrinfo(res);
                    int one = 1;
                    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                    auto* c = new Client();
                    c->fd = fd;
                    return c;
                }
                ::close(fd);
            }
            freeaddrinfo(res);
        }
        if (Clock::now() >= deadline) return nullptr;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}
DTF_EXPORT void dtf_cp_close(void* c) {
    if (!c) return;
    auto* cl = static_cast<Client*>(c);
    if (cl->fd >= 0) ::close(cl->fd);
    delete cl;
}

DTF_EXPORT int dtf_cp_set(void* c, const char* key, const void* data, int64_t len) {
    Writer w; w.str(key); w.blob(data, static_cast<size_t>(len));
    return static_cast<Client*>(c)->call(OP_SET, w);
}
DTF_EXPORT int dtf_cp_get(void* c, const char* key, void* buf, int64_t cap, int64_t* len, int64_t timeout_ms) {
    Writer w; w.str(key); w.i64(timeout_ms);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_GET, w);
    if (st != ST_OK) return st;
    return copy_blob(cl->last, buf, cap, len);
}
DTF_EXPORT int dtf_cp_delete(void* c, const char* key) {
    Writer w; w.str(key);
    return static_cast<Client*>(c)->call(OP_DELETE, w);
}
DTF_EXPORT int dtf_cp_add(void* c, const char* key, int64_t delta, int64_t* out) {
    Writer w; w.str(key); w.i64(delta);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_ADD, w);
    if (st == ST_OK && out) { Reader r(cl->last); *out = r.i64(); }
    return st;
}
DTF_EXPORT int dtf_cp_push(void* c, const char* queue, const void* data, int64_t len) {
    Writer w; w.str(queue); w.blob(data, static_cast<size_t>(len));
    return static_cast<Client*>(c)->call(OP_PUSH, w);
}
DTF_EXPORT int dtf_cp_pop(void* c, const char* queue, void* buf, int64_t cap, int64_t* len, int64_t timeout_ms) {
    Writer w; w.str(queue); w.i64(timeout_ms);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_POP, w);
    if (st != ST_OK) return st;
    return copy_blob(cl->last, buf, cap, len);
}
DTF_EXPORT int dtf_cp_qlen(void* c, const char* queue, int64_t* out) {
    Writer w; w.str(queue);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_QLEN, w);
    if (st == ST_OK && out) { Reader r(cl->last); *out = r.i64(); }
    return st;
}
DTF_EXPORT int dtf_cp_clock_set(void* c, const char* group, int64_t member, int64_t value, int64_t* min_out) {
    Writer w; w.str(group); w.i64(member); w.i64(value);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_CLOCK_SET, w);
    if (st == ST_OK && min_out) { Reader r(cl->last); *min_out = r.i64(); }
    return st;
}
DTF_EXPORT int dtf_cp_clock_min(void* c, const char* group, int64_t* min_out) {
    Writer w; w.str(group);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_CLOCK_MIN, w);
    if (st == ST_OK && min_out) { Reader r(cl->last); *min_out = r.i64(); }
    return st;
}
DTF_EXPORT int dtf_cp_clock_wait(void* c, const char* group, int64_t mine, int64_t bound, int64_t timeout_ms, int64_t* min_out) {
    Writer w; w.str(group); w.i64(mine); w.i64(bound); w.i64(timeout_ms);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_CLOCK_WAIT, w);
    if ((st == ST_OK || st == ST_TIMEOUT) && min_out) { Reader r(cl->last); *min_out = r.i64(); }
    return st;
}
DTF_EXPORT int dtf_cp_barrier(void* c, const char* name, int64_t n, int64_t timeout_ms) {
    Writer w; w.str(name); w.i64(n); w.i64(timeout_ms);
    return static_cast<Client*>(c)->call(OP_BARRIER, w);
}
DTF_EXPORT int dtf_cp_heartbeat(void* c, const char* who) {
    Writer w; w.str(who);
    return static_cast<Client*>(c)->call(OP_HEARTBEAT, w);
}
DTF_EXPORT int dtf_cp_dead(void* c, int64_t max_age_ms, void* buf, int64_t cap, int64_t* len) {
    Writer w; w.i64(max_age_ms);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_DEAD, w);
    if (st != ST_OK) return st;
    return copy_blob(cl->last, buf, cap, len);
}

// ---- folded operations (one round trip each)
// keys: n NUL-terminated strings back to back.  Output: n blobs as u32 length + bytes, back to back, in key order.
DTF_EXPORT int dtf_cp_take_many(void* c, const char* keys, int64_t n, void* buf, int64_t cap, int64_t* len, int64_t timeout_ms) {
    Writer w; w.u32(static_cast<uint32_t>(n));
    const char* p = keys;
    for (int64_t i = 0; i < n; ++i) { std::string k(p); w.str(k); p += k.size() + 1; }
    w.i64(timeout_ms);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_TAKE_MANY, w);
    if (st != ST_OK) return st;
    if (cl->last.size() < 4) return ST_ERROR;
    const int64_t payload = static_cast<int64_t>(cl->last.size()) - 4;
    if (len) *len = payload;
    if (payload > cap) return ST_ERROR;
    std::memcpy(buf, cl->last.data() + 4, static_cast<size_t>(payload));
    return ST_OK;
}
// items: n pairs (queue NUL-terminated, then u32 length + value bytes) back to back.
DTF_EXPORT int dtf_cp_push_many(void* c, const void* items, int64_t items_len, int64_t n) {
    Writer w; w.u32(static_cast<uint32_t>(n));
    const char* p = static_cast<const char*>(items);
    const char* end = p + items_len;
    for (int64_t i = 0; i < n; ++i) {
        std::string q(p); p += q.size() + 1;
        uint32_t vl; if (p + 4 > end) return ST_ERROR; std::memcpy(&vl, p, 4); p += 4;
        if (p + vl > end) return ST_ERROR;
        w.str(q); w.blob(p, vl); p += vl;
    }
    return static_cast<Client*>(c)->call(OP_PUSH_MANY, w);
}
DTF_EXPORT int dtf_cp_clock_set_wait(void* c, const char* group, int64_t member, int64_t value, int64_t bound, int64_t timeout_ms,
                                     int64_t* min_out) {
    Writer w; w.str(group); w.i64(member); w.i64(value); w.i64(bound); w.i64(timeout_ms);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_CLOCK_SET_WAIT, w);
    if ((st == ST_OK || st == ST_TIMEOUT) && min_out) { Reader r(cl->last); *min_out = r.i64(); }
    return st;
}
DTF_EXPORT int dtf_cp_wait_counter(void* c, const char* key, int64_t want, int64_t timeout_ms, int64_t* out) {
    Writer w; w.str(key); w.i64(want); w.i64(timeout_ms);
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_WAIT_COUNTER, w);
    if ((st == ST_OK || st == ST_TIMEOUT) && out) { Reader r(cl->last); *out = r.i64(); }
    return st;
}
DTF_EXPORT int dtf_cp_add_set(void* c, const char* counter, int64_t delta, const char* key, const void* data, int64_t len, int64_t* out) {
    Writer w; w.str(counter ? counter : ""); w.i64(delta); w.str(key); w.blob(data, static_cast<size_t>(len));
    auto* cl = static_cast<Client*>(c);
    int st = cl->call(OP_ADD_SET, w);
    if (st == ST_OK && out) { Reader r(cl->last); *out = r.i64(); }
    return st;
}
