// Multi-threaded stress of the host runtime, meant to be built with -fsanitize=thread:
//   * control plane: one server, several client threads hammering counters, queues, clocks, barriers, blocking GETs
//     and heartbeats concurrently (thread-per-connection server, shared maps behind one mutex);
//   * event writer: several producer threads against the writer's queue thread;
//   * input pipeline: producer threads filling a slot ring against a consumer that releases slots late.
// Prints "STRESS_OK" and exits 0 when every invariant held; ThreadSanitizer reports races on stderr.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
void* dtf_cp_server_start(const char* host, int port);
int dtf_cp_server_port(void* h);
void dtf_cp_server_stop(void* h);
void* dtf_cp_connect(const char* host, int port, double timeout_s);
void dtf_cp_close(void* c);
int dtf_cp_set(void* c, const char* key, const void* data, int64_t len);
int dtf_cp_get(void* c, const char* key, void* buf, int64_t cap, int64_t* len, int64_t timeout_ms);
int dtf_cp_add(void* c, const char* key, int64_t delta, int64_t* out);
int dtf_cp_push(void* c, const char* queue, const void* data, int64_t len);
int dtf_cp_pop(void* c, const char* queue, void* buf, int64_t cap, int64_t* len, int64_t timeout_ms);
int dtf_cp_clock_set(void* c, const char* group, int64_t member, int64_t value, int64_t* min_out);
int dtf_cp_clock_wait(void* c, const char* group, int64_t mine, int64_t bound, int64_t timeout_ms, int64_t* min_out);
int dtf_cp_barrier(void* c, const char* name, int64_t n, int64_t timeout_ms);
int dtf_cp_heartbeat(void* c, const char* who);
void* dtf_events_open(const char* logdir, const char* suffix, int max_queue, double flush_secs);
int dtf_events_add_scalars(void* h, int n, const char* const* tags, const float* values, int64_t step, double wall_time);
int dtf_events_flush(void* h);
int dtf_events_close(void* h);
void* dtf_pipe_create(const void* x, const void* y, int64_t rows, int64_t x_row_bytes, int64_t y_row_bytes, int64_t batch,
                      int shuffle, uint64_t seed, int drop_last, int depth, int threads, void** x_slots, void** y_slots);
int64_t dtf_pipe_start_epoch(void* h, int64_t epoch);
int dtf_pipe_next(void* h, int64_t* slot, int64_t* rows);
void dtf_pipe_release(void* h, int64_t slot);
void dtf_pipe_destroy(void* h);
}

static std::atomic<int> failures{0};
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            failures.fetch_add(1);                                         \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const char* logdir = argc > 1 ? argv[1] : "/tmp";
    const int kThreads = 6, kIters = 200;
    void* server = dtf_cp_server_start("127.0.0.1", 0);
    CHECK(server != nullptr);
    const int port = dtf_cp_server_port(server);

    std::vector<std::thread> threads;
    for (int t = 0; t < kThreads; ++t) {
        threads.emplace_back([=] {
            void* c = dtf_cp_connect("127.0.0.1", port, 10.0);
            CHECK(c != nullptr);
            if (!c) return;
            char who[32];
            std::snprintf(who, sizeof who, "worker:%d", t);
            int64_t v = 0, len = 0, mn = 0;
            for (int i = 0; i < kIters; ++i) {
                CHECK(dtf_cp_add(c, "global_step", 1, &v) == 0);
                int64_t payload = (int64_t)t * 1000000 + i;
                CHECK(dtf_cp_push(c, "tokens", &payload, sizeof payload) == 0);
                int64_t got = 0;
                CHECK(dtf_cp_pop(c, "tokens", &got, sizeof got, &len, 5000) == 0 && len == (int64_t)sizeof got);
                CHECK(dtf_cp_clock_set(c, "ssp", t, i, &mn) == 0);
                CHECK(dtf_cp_clock_wait(c, "ssp", i, 3, 10000, &mn) == 0);   // bounded staleness: nobody runs > 3 ahead
                CHECK(i - mn <= 3);
                if (i % 50 == 0) CHECK(dtf_cp_heartbeat(c, who) == 0);
                if (i % 40 == 0) {
                    char name[32];
                    std::snprintf(name, sizeof name, "b%d", i);
                    CHECK(dtf_cp_barrier(c, name, kThreads, 20000) == 0);
                }
            }
            // blocking GET served by a later SET from another thread
            if (t == 0) {
                int64_t x = 42;
                CHECK(dtf_cp_set(c, "layout", &x, sizeof x) == 0);
            } else {
                int64_t x = 0;
                CHECK(dtf_cp_get(c, "layout", &x, sizeof x, &len, 10000) == 0 && x == 42);
            }
            dtf_cp_close(c);
        });
    }
    for (auto& th : threads) th.join();
    {
        void* c = dtf_cp_connect("127.0.0.1", port, 10.0);
        int64_t v = 0;
        CHECK(c && dtf_cp_add(c, "global_step", 0, &v) == 0 && v == (int64_t)kThreads * kIters);
        dtf_cp_close(c);
    }
    dtf_cp_server_stop(server);

    void* w = dtf_events_open(logdir, ".stress", 64, 0.05);
    CHECK(w != nullptr);
    threads.clear();
    for (int t = 0; t < 4; ++t) {
        threads.emplace_back([=] {
            const char* tags[2] = {"loss", "accuracy"};
            for (int i = 0; i < 300; ++i) {
                const float vals[2] = {1.0f / (i + 1), (float)t};
                CHECK(dtf_events_add_scalars(w, 2, tags, vals, i, 0.0) == 0);
                if (i % 100 == 0) dtf_events_flush(w);
            }
        });
    }
    for (auto& th : threads) th.join();
    CHECK(dtf_events_close(w) == 0);

    // input pipeline: 4 producers, 3 slots, a consumer that releases late, one epoch abandoned half way
    {
        const int64_t rows = 1003, batch = 32, depth = 3;
        std::vector<int64_t> x(rows), y(rows);
        for (int64_t i = 0; i < rows; ++i) { x[i] = i; y[i] = 7 * i; }
        std::vector<std::vector<int64_t>> xs(depth, std::vector<int64_t>(batch)), ys(depth, std::vector<int64_t>(batch));
        std::vector<void*> xp, yp;
        for (int i = 0; i < depth; ++i) { xp.push_back(xs[i].data()); yp.push_back(ys[i].data()); }
        void* pipe = dtf_pipe_create(x.data(), y.data(), rows, 8, 8, batch, 1, 11, 0, (int)depth, 4, xp.data(), yp.data());
        CHECK(pipe != nullptr);
        for (int epoch = 0; pipe && epoch < 4; ++epoch) {
            const int64_t nb = dtf_pipe_start_epoch(pipe, epoch);
            CHECK(nb == (rows + batch - 1) / batch);
            std::vector<int> seen(rows, 0);
            int64_t slot = 0, n = 0, prev = -1, got = 0;
            while (dtf_pipe_next(pipe, &slot, &n) == 0) {
                for (int64_t i = 0; i < n; ++i) {
                    const int64_t r = xs[slot][i];
                    CHECK(r >= 0 && r < rows && ys[slot][i] == 7 * r);
                    if (r >= 0 && r < rows) ++seen[r];
                }
                if (prev >= 0) dtf_pipe_release(pipe, prev);     // hold one slot while the next is read
                prev = slot;
                if (epoch == 2 && ++got == nb / 2) break;          // walk away in the middle of an epoch
            }
            if (prev >= 0) dtf_pipe_release(pipe, prev);
            if (epoch != 2)
                for (int64_t i = 0; i < rows; ++i) CHECK(seen[i] == 1);
        }
        if (pipe) dtf_pipe_destroy(pipe);
    }

    if (failures.load() == 0) std::puts("STRESS_OK");
    return failures.load() == 0 ? 0 : 1;
}
