// Host input pipeline: shuffled mini-batches of a host-resident data set are gathered by worker threads into a ring
// of caller-provided staging slots (pinned memory, so the device copy that follows is asynchronous), a few batches
// ahead of the consumer.  It replaces the per-step feed of Python lists in the reference
// (/root/reference/example.py:209-213: every sess.run converts and ships a slice of a list of lists).
//
//   batch b of an epoch   -> slot b % depth
//   producer threads      claim batch numbers in order, wait until the slot's previous tenant was released, gather
//                         rows perm[b*batch .. ) of x and y into it, mark it ready
//   consumer (next)       waits for batch c to be ready, gets (slot, rows); release(slot) hands the slot back once
//                         the copy out of it has completed
//
// The permutation of epoch e depends on (seed, e) only: every run, and every restart from a checkpoint, sees the
// same order.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define DTF_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Pipe {
    const uint8_t* x = nullptr;
    const uint8_t* y = nullptr;
    int64_t rows = 0, x_row = 0, y_row = 0, batch = 0;
    bool shuffle = false, drop_last = false;
    uint64_t seed = 0;
    int depth = 0;
    std::vector<uint8_t*> xs, ys;

    std::mutex mu;
    std::condition_variable cv;
    std::vector<int64_t> perm;
    int64_t nbatches = 0;          // batches of the running epoch
    int64_t next_claim = 0;        // next batch a producer takes
    int64_t next_consume = 0;      // next batch next() returns
    std::vector<int64_t> ready;    // per slot: batch number it holds and is ready, or -1
    std::vector<int64_t> free_for; // per slot: the batch number that may fill it next
    uint64_t generation = 0;       // bumps at every start_epoch: stale producers drop their work
    int inflight = 0;              // producers copying rows right now
    bool stopping = false;
    std::vector<std::thread> workers;

    void work() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stopping || next_claim < nbatches; });
            if (stopping) return;
            const int64_t b = next_claim++;
            const uint64_t gen = generation;
            const int slot = static_cast<int>(b % depth);
            cv.wait(lk, [&] { return stopping || generation != gen || free_for[slot] == b; });
            if (stopping) return;
            if (generation != gen) continue;
            const int64_t lo = b * batch;
            const int64_t n = std::min<int64_t>(batch, rows - lo);
            ++inflight;
            lk.unlock();
            uint8_t* dx = xs[slot];
            uint8_t* dy = ys.empty() ? nullptr : ys[slot];
            for (int64_t i = 0; i < n; ++i) {
                const int64_t src = shuffle ? perm[lo + i] : lo + i;
                std::memcpy(dx + i * x_row, x + src * x_row, static_cast<size_t>(x_row));
                if (dy) std::memcpy(dy + i * y_row, y + src * y_row, static_cast<size_t>(y_row));
            }
            lk.lock();
            --inflight;
            if (generation == gen) ready[slot] = b;
            cv.notify_all();
        }
    }
};

}  // namespace

DTF_EXPORT void* dtf_pipe_create(const void* x, const void* y, int64_t rows, int64_t x_row_bytes, int64_t y_row_bytes,
                                 int64_t batch, int shuffle, uint64_t seed, int drop_last, int depth, int threads,
                                 void** x_slots, void** y_slots) {
    if (!x || rows <= 0 || x_row_bytes <= 0 || batch <= 0 || depth < 2 || threads < 1 || !x_slots) return nullptr;
    if (y && (!y_slots || y_row_bytes <= 0)) return nullptr;
    auto* p = new Pipe();
    p->x = static_cast<const uint8_t*>(x);
    p->y = static_cast<const uint8_t*>(y);
    p->rows = rows; p->x_row = x_row_bytes; p->y_row = y_row_bytes; p->batch = batch;
    p->shuffle = shuffle != 0; p->drop_last = drop_last != 0; p->seed = seed; p->depth = depth;
    for (int i = 0; i < depth; ++i) {
        p->xs.push_back(static_cast<uint8_t*>(x_slots[i]));
        if (y) p->ys.push_back(static_cast<uint8_t*>(y_slots[i]));
    }
    p->ready.assign(depth, -1);
    p->free_for.assign(depth, 0);
    for (int t = 0; t < threads; ++t) p->workers.emplace_back([p] { p->work(); });
    return p;
}

// Begin epoch `epoch`: returns its number of batches.  Slots still held by the consumer must have been released.
DTF_EXPORT int64_t dtf_pipe_start_epoch(void* h, int64_t epoch) {
    auto* p = static_cast<Pipe*>(h);
    std::vector<int64_t> perm;
    if (p->shuffle) {
        perm.resize(static_cast<size_t>(p->rows));
        for (int64_t i = 0; i < p->rows; ++i) perm[i] = i;
        uint64_t s = p->seed * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(epoch) + 1;
        for (int64_t i = p->rows - 1; i > 0; --i) {
            const int64_t j = static_cast<int64_t>(splitmix64(s) % static_cast<uint64_t>(i + 1));
            std::swap(perm[i], perm[j]);
        }
    }
    std::unique_lock<std::mutex> lk(p->mu);
    ++p->generation;                                   // producers of an abandoned epoch drop what they hold ...
    p->nbatches = 0;
    p->cv.notify_all();
    p->cv.wait(lk, [&] { return p->inflight == 0; });   // ... and nobody is still writing into a slot
    p->perm.swap(perm);
    p->nbatches = p->drop_last ? p->rows / p->batch : (p->rows + p->batch - 1) / p->batch;
    p->next_claim = 0;
    p->next_consume = 0;
    for (int i = 0; i < p->depth; ++i) { p->ready[i] = -1; p->free_for[i] = i; }
    p->cv.notify_all();
    return p->nbatches;
}

// 0: *slot holds *rows rows of the next batch; 1: the epoch is over; -1: the pipeline is shutting down
DTF_EXPORT int dtf_pipe_next(void* h, int64_t* slot, int64_t* rows) {
    auto* p = static_cast<Pipe*>(h);
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->next_consume >= p->nbatches) return 1;
    const int64_t b = p->next_consume;
    const int s = static_cast<int>(b % p->depth);
    p->cv.wait(lk, [&] { return p->stopping || p->ready[s] == b; });
    if (p->stopping) return -1;
    ++p->next_consume;
    *slot = s;
    *rows = std::min<int64_t>(p->batch, p->rows - b * p->batch);
    return 0;
}

// The consumer is done with the slot's contents (its device copy has completed): batch held + depth may fill it.
DTF_EXPORT void dtf_pipe_release(void* h, int64_t slot) {
    auto* p = static_cast<Pipe*>(h);
    std::lock_guard<std::mutex> lk(p->mu);
    if (slot < 0 || slot >= p->depth || p->ready[slot] < 0) return;
    p->free_for[slot] = p->ready[slot] + p->depth;
    p->ready[slot] = -1;
    p->cv.notify_all();
}

// Row index that position `i` of epoch's order maps to (for tests and for resuming inside an epoch).
DTF_EXPORT int64_t dtf_pipe_perm_at(void* h, int64_t i) {
    auto* p = static_cast<Pipe*>(h);
    std::lock_guard<std::mutex> lk(p->mu);
    if (i < 0 || i >= p->rows) return -1;
    return p->shuffle ? p->perm[static_cast<size_t>(i)] : i;
}

DTF_EXPORT void dtf_pipe_destroy(void* h) {
    auto* p = static_cast<Pipe*>(h);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stopping = true;
        p->cv.notify_all();
    }
    for (auto& t : p->workers) t.join();
    delete p;
}
