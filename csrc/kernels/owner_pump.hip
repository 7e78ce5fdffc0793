// Owner pump: the native service loop of a parameter-server owner, and the worker-side doorbell.
//
// In the reference a parameter-server task applies every gradient a worker sends as it arrives, lock-free, and the
// workers never wait for one another (/root/reference/example.py:136-139,168-170).  Here an owner is a thread of this
// file plus one HIP stream: it watches the doorbell words of its workers, launches the fused apply-gradients kernel on
// the gradient shard a worker has put into its inbox, and acknowledges through the same words.  Nothing on this path
// crosses a socket or waits for a stream on the host:
//
//   worker w, step t                                      owner j
//   ----------------                                      -------
//   put grad shard -> inbox[w][t % D]   (stream copy)
//   host function on that stream:                         pump thread sees lane[w].door > consumed[w]
//       lane[w].pull_version[t % D] = version pulled        apply kernel on the owner's stream (var, m, v in its HBM)
//       lane[w].door = t + 1                                host function on that stream:
//                                                               lane[w].ack_version[t % D] = version after the apply
//   (before reusing inbox[w][(t+D) % D]:                        ctl.version, ctl.global_step, lane[w].ack = t + 1
//    wait until lane[w].ack > t)
//
// The words live in host memory shared by all processes of the node (one control block per owner); they are only
// ever written by host code - a stream's host function runs after everything queued before it has completed, which
// is the ordering and visibility the HIP runtime guarantees for copies and kernels.  The gradient and parameter data
// itself stays in device memory (peer-mapped regions, see peer_mailbox.hip).
//
// On a CPU (the plumbing configuration and the tests) the same thread applies the update itself and the doorbell is
// a plain store.
#include "common.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

extern "C" int dtf_adam_apply(float* var, float* m, float* v, const float* g, long long n, float lr, float b1, float b2,
                              float eps, float grad_scale, const long long* step_ptr, long long* advance_step,
                              float* beta_powers, unsigned int* ticket, hipStream_t stream);
extern "C" int dtf_sgd_apply(float* var, float* mom, const float* g, long long n, float lr, float momentum,
                             float grad_scale, long long* advance_step, unsigned int* ticket, hipStream_t stream);

namespace {

constexpr int PUMP_MAX_DEPTH = 4;
constexpr int PUMP_HIST = 64;

// One per (owner, worker).  128 bytes: the worker-written and the owner-written halves sit in different cache lines.
struct Lane {
    std::atomic<uint64_t> door;                 // gradients of this worker complete in the owner's inbox   [worker]
    std::atomic<uint64_t> clock;                // the worker's clock for the staleness bound               [worker]
    uint64_t pull_version[PUMP_MAX_DEPTH];      // owner version the gradient in slot s was computed from   [worker]
    uint64_t pad0[2];
    std::atomic<uint64_t> ack;                  // gradients of this worker applied                         [owner]
    uint64_t ack_version[PUMP_MAX_DEPTH];       // owner version right after the apply of slot s            [owner]
    uint64_t pad1[3];
};
static_assert(sizeof(Lane) == 128, "lane layout is shared with the Python side");

struct Ctl {
    std::atomic<uint64_t> version;              // updates applied by this owner
    std::atomic<uint64_t> global_step;          // owner 0: the cluster-wide step (step at initialisation + version)
    std::atomic<uint64_t> gs_base;              // global step when the owner was (re-)initialised
    uint64_t pad[13];
    Lane lane[1];                               // [workers]
};
static_assert(offsetof(Ctl, lane) == 128, "control block layout is shared with the Python side");

struct Config {
    int on_gpu, owner, workers, depth;
    int optimizer;                              // 0 Adam, 1 SGD (momentum slot optional)
    int pad;
    long long n;                                // elements of this owner's shard
    long long slot_stride;                      // elements between inbox slots
    float *var, *m, *v, *inbox;
    long long* step;                            // device (GPU) or host (CPU) int64: updates applied, for bias correction
    float* beta_powers;
    unsigned int* ticket;
    float lr, b1, b2, eps, momentum;
    int pad2;
    void* ctl;
};

struct Pump;

struct AckRecord {
    Pump* pump;
    int worker, slot;
    uint64_t seq, version, pulled;
};

struct Pump {
    Config cfg;
    Ctl* ctl;
    hipStream_t stream = nullptr;
    std::thread thread;
    std::atomic<int> stop{0}, pause_req{0}, paused{0}, error{0};
    std::vector<uint64_t> consumed;
    uint64_t enqueued_version = 0;              // version after everything launched so far
    std::atomic<uint64_t> hist[PUMP_HIST];
    std::vector<AckRecord> ring;                // records of acknowledgements in flight
    size_t ring_pos = 0;
};

void acknowledge(const AckRecord& r) {
    Pump* p = r.pump;
    Lane& lane = p->ctl->lane[r.worker];
    lane.ack_version[r.slot] = r.version;
    p->ctl->version.store(r.version, std::memory_order_release);
    if (p->cfg.owner == 0)
        p->ctl->global_step.store(p->ctl->gs_base.load(std::memory_order_relaxed) + r.version, std::memory_order_release);
    const uint64_t before = r.version - 1;
    uint64_t st = before > r.pulled ? before - r.pulled : 0;
    if (st >= (uint64_t)PUMP_HIST) st = PUMP_HIST - 1;
    p->hist[st].fetch_add(1, std::memory_order_relaxed);
    lane.ack.store(r.seq, std::memory_order_release);
}

void ack_host_fn(void* data) { acknowledge(*static_cast<AckRecord*>(data)); }

// The TensorFlow ApplyAdam formula (epsilon outside the bias correction), as optim.hip computes it.
void cpu_apply(const Config& c, const float* g) {
    const long long n = c.n;
    if (c.optimizer == 0) {
        const float t = (float)(*c.step + 1);
        const float lr_t = c.lr * std::sqrt(1.0f - std::pow(c.b2, t)) / (1.0f - std::pow(c.b1, t));
        const float omb1 = 1.0f - c.b1, omb2 = 1.0f - c.b2;
        for (long long i = 0; i < n; ++i) {
            const float gi = g[i];
            const float mi = c.b1 * c.m[i] + omb1 * gi;
            const float vi = c.b2 * c.v[i] + omb2 * gi * gi;
            c.m[i] = mi;
            c.v[i] = vi;
            c.var[i] -= lr_t * mi / (std::sqrt(vi) + c.eps);
        }
        if (c.beta_powers) {
            c.beta_powers[0] *= c.b1;
            c.beta_powers[1] *= c.b2;
        }
    } else {
        for (long long i = 0; i < n; ++i) {
            float gi = g[i];
            if (c.m && c.momentum > 0.0f) {
                gi = c.momentum * c.m[i] + gi;
                c.m[i] = gi;
            }
            c.var[i] -= c.lr * gi;
        }
    }
    *c.step += 1;
}

int gpu_apply(Pump* p, const float* g) {
    const Config& c = p->cfg;
    if (c.n <= 0) return 0;
    if (c.optimizer == 0)
        return dtf_adam_apply(c.var, c.m, c.v, g, c.n, c.lr, c.b1, c.b2, c.eps, 1.0f, c.step, c.step, c.beta_powers,
                              c.ticket, p->stream);
    return dtf_sgd_apply(c.var, c.momentum > 0.0f ? c.m : nullptr, g, c.n, c.lr, c.momentum, 1.0f, c.step, c.ticket,
                         p->stream);
}

void pump_loop(Pump* p, int device) {
    const Config& c = p->cfg;
    if (c.on_gpu && hipSetDevice(device) != hipSuccess) {
        p->error.store(-10);
        return;
    }
    int idle = 0, first = 0;
    while (!p->stop.load(std::memory_order_acquire)) {
        if (p->pause_req.load(std::memory_order_acquire)) {
            if (c.on_gpu && hipStreamSynchronize(p->stream) != hipSuccess) p->error.store(-11);
            p->paused.store(1, std::memory_order_release);
            while (p->pause_req.load(std::memory_order_acquire) && !p->stop.load(std::memory_order_acquire))
                std::this_thread::sleep_for(std::chrono::microseconds(100));
            p->paused.store(0, std::memory_order_release);
            continue;
        }
        bool any = false;
        for (int k = 0; k < c.workers; ++k) {
            const int w = (first + k) % c.workers;          // rotate the starting lane: no worker is always served last
            Lane& lane = p->ctl->lane[w];
            const uint64_t door = lane.door.load(std::memory_order_acquire);
            if (door <= p->consumed[w]) continue;
            const uint64_t seq = p->consumed[w];
            const int slot = (int)(seq % (uint64_t)c.depth);
            const float* g = c.inbox + ((long long)w * c.depth + slot) * c.slot_stride;
            AckRecord rec{p, w, slot, seq + 1, ++p->enqueued_version, lane.pull_version[slot]};
            if (c.on_gpu) {
                int rc = gpu_apply(p, g);
                AckRecord* slot_rec = &p->ring[p->ring_pos];
                p->ring_pos = (p->ring_pos + 1) % p->ring.size();
                *slot_rec = rec;
                if (rc == 0) rc = (int)hipLaunchHostFunc(p->stream, ack_host_fn, slot_rec);
                if (rc != 0) {
                    p->error.store(rc);
                    return;
                }
            } else {
                if (c.n > 0) cpu_apply(c, g);
                acknowledge(rec);
            }
            p->consumed[w] = seq + 1;
            any = true;
        }
        first = (first + 1) % (c.workers > 0 ? c.workers : 1);
        if (any) {
            idle = 0;
        } else if (++idle < 200) {
            __builtin_ia32_pause();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(idle < 2000 ? 20 : 200));
        }
    }
    if (c.on_gpu) (void)hipStreamSynchronize(p->stream);
}

struct RingRecord {
    int count, slot;
    uint64_t seq, clock;
    Lane* lanes[64];
    uint64_t pulled[64];
};

void ring_now(const RingRecord& r) {
    for (int i = 0; i < r.count; ++i) {
        Lane* lane = r.lanes[i];
        lane->pull_version[r.slot] = r.pulled[i];
        if (i == 0) lane->clock.store(r.clock, std::memory_order_release);   // lanes[0] belongs to owner 0's block
        lane->door.store(r.seq, std::memory_order_release);
    }
}

void ring_host_fn(void* data) {
    RingRecord* r = static_cast<RingRecord*>(data);
    ring_now(*r);
    delete r;
}

}  // namespace

DTF_API int dtf_pump_config_bytes() { return (int)sizeof(Config); }
DTF_API int dtf_pump_lane_bytes() { return (int)sizeof(Lane); }
DTF_API long long dtf_pump_ctl_bytes(int workers) { return (long long)(offsetof(Ctl, lane) + sizeof(Lane) * (size_t)workers); }

// Starts the owner's service thread (and, on a GPU, its stream) for the shard described by `cfg`.  The control block
// must have been zeroed by the caller; `consumed` starts at the doorbell values found there.
DTF_API void* dtf_pump_start(const void* cfg_bytes) {
    Config cfg;
    std::memcpy(&cfg, cfg_bytes, sizeof cfg);
    if (cfg.workers < 1 || cfg.workers > 64 || cfg.depth < 1 || cfg.depth > PUMP_MAX_DEPTH || !cfg.ctl || cfg.n < 0)
        return nullptr;
    if (cfg.n > 0 && (!cfg.var || !cfg.inbox || !cfg.step)) return nullptr;
    auto* p = new Pump();
    p->cfg = cfg;
    p->ctl = static_cast<Ctl*>(cfg.ctl);
    p->consumed.resize(cfg.workers);
    for (int w = 0; w < cfg.workers; ++w) p->consumed[w] = p->ctl->lane[w].ack.load();
    p->enqueued_version = p->ctl->version.load();
    for (auto& h : p->hist) h.store(0);
    // every worker has at most `depth` gradients outstanding, so workers * depth records are never all in flight
    p->ring.resize((size_t)cfg.workers * cfg.depth + 8);
    int device = 0;
    if (cfg.on_gpu) {
        if (hipGetDevice(&device) != hipSuccess) { delete p; return nullptr; }
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            delete p;
            return nullptr;
        }
    }
    p->thread = std::thread(pump_loop, p, device);
    return p;
}

// Stops applying (everything already launched completes) / resumes.  While paused the owner's shard is stable.
DTF_API int dtf_pump_pause(void* h) {
    auto* p = static_cast<Pump*>(h);
    p->pause_req.store(1, std::memory_order_release);
    for (int i = 0; i < 600000 && !p->paused.load(std::memory_order_acquire); ++i) {
        if (p->error.load()) return p->error.load();
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    return p->paused.load() ? 0 : -12;
}

DTF_API int dtf_pump_resume(void* h, int resync) {
    auto* p = static_cast<Pump*>(h);
    if (resync) {
        // the control block was reset while paused (owner re-initialised): start over from what it says now
        for (int w = 0; w < p->cfg.workers; ++w) p->consumed[w] = p->ctl->lane[w].ack.load();
        p->enqueued_version = p->ctl->version.load();
    }
    p->pause_req.store(0, std::memory_order_release);
    for (int i = 0; i < 600000 && p->paused.load(std::memory_order_acquire); ++i)
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    return 0;
}

DTF_API int dtf_pump_error(void* h) { return static_cast<Pump*>(h)->error.load(); }

DTF_API void* dtf_pump_stream(void* h) { return static_cast<Pump*>(h)->stream; }

DTF_API void dtf_pump_histogram(void* h, long long* out) {
    auto* p = static_cast<Pump*>(h);
    for (int i = 0; i < PUMP_HIST; ++i) out[i] = (long long)p->hist[i].load();
}

DTF_API int dtf_pump_stop(void* h) {
    auto* p = static_cast<Pump*>(h);
    p->stop.store(1, std::memory_order_release);
    if (p->thread.joinable()) p->thread.join();
    const int err = p->error.load();
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return err;
}

// Worker side: after everything queued on `stream` so far (the puts of this step's gradient shards), tell `count`
// owners that gradient number `seq` (1-based) of this worker is complete in slot `slot` of their inboxes.
// `lanes[i]` is this worker's lane in owner i's control block, `pulled[i]` the owner version the gradient was
// computed from.  `clock` is published in lanes[0] (owner 0's block) for the staleness bound.  stream == null on a
// CPU: the stores happen now.
DTF_API int dtf_doorbell_ring(hipStream_t stream, int on_gpu, void* const* lanes, const unsigned long long* pulled, int count,
                              int slot, unsigned long long seq, unsigned long long clock) {
    if (count < 1 || count > 64 || slot < 0 || slot >= PUMP_MAX_DEPTH) return -2;
    auto* r = new RingRecord();
    r->count = count;
    r->slot = slot;
    r->seq = seq;
    r->clock = clock;
    for (int i = 0; i < count; ++i) {
        r->lanes[i] = static_cast<Lane*>(lanes[i]);
        r->pulled[i] = pulled[i];
    }
    if (!on_gpu) {
        ring_now(*r);
        delete r;
        return 0;
    }
    const hipError_t e = hipLaunchHostFunc(stream, ring_host_fn, r);
    if (e != hipSuccess) {
        delete r;
        return (int)e;
    }
    return 0;
}
