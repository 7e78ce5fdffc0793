// TensorBoard event files.
//
// What tf.summary.FileWriter / the Keras TensorBoard callback produce for
// /root/reference/example.py:172-174,195,219 and /root/reference/example2.py:197:
//   file   events.out.tfevents.<unix seconds>.<hostname>[suffix]
//   record u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)
//   payload = serialized Event { wall_time = 1 (double), step = 2 (int64),
//                                file_version = 3 | graph_def = 4 | summary = 5 }
//   Summary { repeated Value value = 1 { tag = 1, simple_value = 2 (float) | histo = 5 } }
// The first record is Event{file_version: "brain.Event:2"}.  Records go through a queue to a
// writer thread (flushed every flush_secs, on flush() and on close()), so the training loop
// never blocks on the file system.
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "crc32c.h"

namespace {

void put_varint(std::string* s, uint64_t v) {
    while (v >= 128) {
        s->push_back(static_cast<char>((v & 0x7f) | 0x80));
        v >>= 7;
    }
    s->push_back(static_cast<char>(v));
}
void put_tag(std::string* s, int field, int wire) { put_varint(s, static_cast<uint64_t>((field << 3) | wire)); }
void put_double(std::string* s, int field, double v) {
    put_tag(s, field, 1);
    s->append(reinterpret_cast<const char*>(&v), 8);
}
void put_float(std::string* s, int field, float v) {
    put_tag(s, field, 5);
    s->append(reinterpret_cast<const char*>(&v), 4);
}
void put_int64(std::string* s, int field, int64_t v) {
    put_tag(s, field, 0);
    put_varint(s, static_cast<uint64_t>(v));
}
void put_bytes(std::string* s, int field, const void* p, size_t n) {
    put_tag(s, field, 2);
    put_varint(s, n);
    s->append(static_cast<const char*>(p), n);
}
void put_bytes(std::string* s, int field, const std::string& b) { put_bytes(s, field, b.data(), b.size()); }

std::string event_header(double wall, int64_t step) {
    std::string e;
    put_double(&e, 1, wall);
    if (step != 0) put_int64(&e, 2, step);
    return e;
}

double now_seconds() {
    return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

bool mkdirs(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && cur != "/") {
                if (::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
            }
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
    return true;
}

struct EventsWriter {
    std::string path;
    FILE* file = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::string> queue;
    size_t max_queue = 10;
    double flush_secs = 120.0;
    bool closing = false, flush_requested = false;
    uint64_t written = 0, flushed_upto = 0, enqueued = 0;
    std::thread worker;

    void write_record(const std::string& payload) {
        const uint64_t len = payload.size();
        const uint32_t len_crc = dtf::crc32c_mask(dtf::crc32c(&len, 8));
        const uint32_t data_crc = dtf::crc32c_mask(dtf::crc32c(payload.data(), payload.size()));
        std::fwrite(&len, 8, 1, file);
        std::fwrite(&len_crc, 4, 1, file);
        if (len) std::fwrite(payload.data(), 1, payload.size(), file);
        std::fwrite(&data_crc, 4, 1, file);
    }

    void run() {
        auto last_flush = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            // system-clock wait (pthread_cond_timedwait): a 200 ms slice, and visible to ThreadSanitizer
            cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(200),
                          [&] { return closing || flush_requested || !queue.empty(); });
            std::deque<std::string> batch;
            batch.swap(queue);
            const bool do_flush = flush_requested || closing ||
                                  std::chrono::duration<double>(std::chrono::steady_clock::now() - last_flush).count() >= flush_secs;
            const bool stop = closing;
            lk.unlock();
            for (auto& r : batch) write_record(r);
            if (do_flush) {
                std::fflush(file);
                last_flush = std::chrono::steady_clock::now();
            }
            lk.lock();
            written += batch.size();
            if (do_flush) {
                flushed_upto = written;
                flush_requested = false;
            }
            cv.notify_all();
            if (stop && queue.empty()) break;
        }
    }

    bool open(const std::string& logdir, const std::string& suffix, int max_q, double flush_s) {
        if (!mkdirs(logdir)) return false;
        char host[256] = "localhost";
        ::gethostname(host, sizeof(host) - 1);
        char name[512];
        std::snprintf(name, sizeof(name), "%s/events.out.tfevents.%010lld.%s%s", logdir.c_str(),
                      static_cast<long long>(std::time(nullptr)), host, suffix.c_str());
        path = name;
        // several writers may open in the same second (the script's own and the session hooks'):
        // never overwrite, append ".<pid>.<n>" instead
        for (int n = 0; ::access(path.c_str(), F_OK) == 0; ++n)
            path = std::string(name) + "." + std::to_string(::getpid()) + "." + std::to_string(n);
        file = std::fopen(path.c_str(), "wb");
        if (!file) return false;
        max_queue = max_q > 0 ? static_cast<size_t>(max_q) : 10;
        flush_secs = flush_s > 0 ? flush_s : 120.0;
        std::string e = event_header(now_seconds(), 0);
        put_bytes(&e, 3, std::string("brain.Event:2"));
        write_record(e);
        std::fflush(file);
        worker = std::thread([this] { run(); });
        return true;
    }

    void add(std::string&& payload) {
        std::unique_lock<std::mutex> lk(mu);
        // bounded queue: the producer waits only if the writer thread is max_queue records behind
        cv.wait(lk, [&] { return queue.size() < max_queue * 64 || closing; });
        queue.push_back(std::move(payload));
        ++enqueued;
        cv.notify_all();
    }

    void flush() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t target = enqueued;
        flush_requested = true;
        cv.notify_all();
        cv.wait(lk, [&] { return flushed_upto >= target || closing; });
    }

    void close() {
        {
            std::lock_guard<std::mutex> lk(mu);
            closing = true;
            cv.notify_all();
        }
        if (worker.joinable()) worker.join();
        if (file) {
            std::fflush(file);
            std::fclose(file);
            file = nullptr;
        }
    }
};

}  // namespace

DTF_EXPORT void* dtf_events_open(const char* logdir, const char* suffix, int max_queue, double flush_secs) {
    auto* w = new EventsWriter();
    if (!w->open(logdir, suffix ? suffix : "", max_queue, flush_secs)) {
        delete w;
        return nullptr;
    }
    return w;
}

DTF_EXPORT int dtf_events_path(void* h, char* buf, int cap) {
    auto* w = static_cast<EventsWriter*>(h);
    if (static_cast<int>(w->path.size()) + 1 > cap) return -1;
    std::memcpy(buf, w->path.c_str(), w->path.size() + 1);
    return 0;
}

// One Event carrying n scalar values (tags are NUL-terminated strings); wall_time <= 0 means "now".
DTF_EXPORT int dtf_events_add_scalars(void* h, int n, const char* const* tags, const float* values, int64_t step,
                                      double wall_time) {
    std::string summary;
    for (int i = 0; i < n; ++i) {
        std::string v;
        put_bytes(&v, 1, tags[i], std::strlen(tags[i]));
        put_float(&v, 2, values[i]);
        put_bytes(&summary, 1, v);
    }
    std::string e = event_header(wall_time > 0 ? wall_time : now_seconds(), step);
    put_bytes(&e, 5, summary);
    static_cast<EventsWriter*>(h)->add(std::move(e));
    return 0;
}

// Event{summary: <already serialized Summary proto>}
DTF_EXPORT int dtf_events_add_summary(void* h, const void* summary, int64_t len, int64_t step, double wall_time) {
    std::string e = event_header(wall_time > 0 ? wall_time : now_seconds(), step);
    put_bytes(&e, 5, summary, static_cast<size_t>(len));
    static_cast<EventsWriter*>(h)->add(std::move(e));
    return 0;
}

DTF_EXPORT int dtf_events_add_graph(void* h, const void* graph, int64_t len, double wall_time) {
    std::string e = event_header(wall_time > 0 ? wall_time : now_seconds(), 0);
    put_bytes(&e, 4, graph, static_cast<size_t>(len));
    static_cast<EventsWriter*>(h)->add(std::move(e));
    return 0;
}

// HistogramProto { min=1 max=2 num=3 sum=4 sum_squares=5 bucket_limit=6 (packed) bucket=7 (packed) }
DTF_EXPORT int dtf_events_add_histogram(void* h, const char* tag, int64_t step, double wall_time, double mn, double mx,
                                        double num, double sum, double sum_squares, const double* limits,
                                        const double* counts, int nbuckets) {
    std::string hp;
    put_double(&hp, 1, mn);
    put_double(&hp, 2, mx);
    put_double(&hp, 3, num);
    put_double(&hp, 4, sum);
    put_double(&hp, 5, sum_squares);
    put_bytes(&hp, 6, limits, sizeof(double) * static_cast<size_t>(nbuckets));
    put_bytes(&hp, 7, counts, sizeof(double) * static_cast<size_t>(nbuckets));
    std::string v;
    put_bytes(&v, 1, tag, std::strlen(tag));
    put_bytes(&v, 5, hp);
    std::string summary;
    put_bytes(&summary, 1, v);
    std::string e = event_header(wall_time > 0 ? wall_time : now_seconds(), step);
    put_bytes(&e, 5, summary);
    static_cast<EventsWriter*>(h)->add(std::move(e));
    return 0;
}

DTF_EXPORT int dtf_events_flush(void* h) {
    static_cast<EventsWriter*>(h)->flush();
    return 0;
}

DTF_EXPORT int dtf_events_close(void* h) {
    auto* w = static_cast<EventsWriter*>(h);
    w->close();
    delete w;
    return 0;
}
