// Feeding a session-style loop: `sess.run(train_op, {x: batch, y: labels})` hands the runtime two small host arrays
// per step (/root/reference/example.py:207-213 - batches of 50 rows).  Each goes host -> pinned staging buffer ->
// device with one asynchronous copy on the step's stream; an event per staging buffer says when that copy has been
// executed, and the buffer is only rewritten after it.  All of it - the wait, the memcpy into the pinned buffer, the
// copy and the event record - is one native call, a few microseconds on the CPU: the same sequence issued through
// the tensor library costs ~20 us per fed value, which is a quarter of a batch-50 training step.
#include "common.h"

#include <cstring>

namespace {

struct FeedRing {
    int depth = 0;
    size_t bytes = 0;
    int next = 0;
    void** host = nullptr;      // pinned, owned by the caller (the tensor library keeps them alive)
    void** dev = nullptr;       // device, owned by the caller
    hipEvent_t* done = nullptr;
};

}  // namespace

// `host[i]` / `dev[i]`: `depth` pinned host buffers and device buffers of `bytes` each.
DTF_API int dtf_feed_ring_create(int depth, long long bytes, void** host, void** dev, void** out) {
    if (depth < 1 || bytes < 1 || !host || !dev || !out) return -1;
    FeedRing* r = new FeedRing;
    r->depth = depth;
    r->bytes = (size_t)bytes;
    r->host = new void*[depth];
    r->dev = new void*[depth];
    r->done = new hipEvent_t[depth];
    for (int i = 0; i < depth; ++i) {
        r->host[i] = host[i];
        r->dev[i] = dev[i];
        hipError_t e = hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int k = 0; k < i; ++k) (void)hipEventDestroy(r->done[k]);
            delete[] r->host; delete[] r->dev; delete[] r->done; delete r;
            return (int)e;
        }
    }
    *out = r;
    return 0;
}

DTF_API int dtf_feed_ring_destroy(void* handle) {
    FeedRing* r = static_cast<FeedRing*>(handle);
    if (!r) return 0;
    for (int i = 0; i < r->depth; ++i) (void)hipEventDestroy(r->done[i]);
    delete[] r->host; delete[] r->dev; delete[] r->done; delete r;
    return 0;
}

// The slot the next `dtf_feed_ring_push` will use, once its previous copy has been executed: the caller may write the
// pinned buffer itself (a dtype conversion) and then push with `src == nullptr`.
DTF_API int dtf_feed_ring_acquire(void* handle, int* slot) {
    FeedRing* r = static_cast<FeedRing*>(handle);
    if (!r || !slot) return -1;
    hipError_t e = hipEventSynchronize(r->done[r->next]);      // returns at once for an event never recorded
    if (e != hipSuccess) return (int)e;
    *slot = r->next;
    return 0;
}

// `src` (host, `bytes` long; nullptr: the pinned buffer already holds the data) -> pinned -> device, on `stream`.
DTF_API int dtf_feed_ring_push(void* handle, const void* src, void* stream, int* slot) {
    FeedRing* r = static_cast<FeedRing*>(handle);
    if (!r || !slot) return -1;
    const int i = r->next;
    hipError_t e = hipEventSynchronize(r->done[i]);
    if (e != hipSuccess) return (int)e;
    if (src) std::memcpy(r->host[i], src, r->bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    e = hipMemcpyAsync(r->dev[i], r->host[i], r->bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    e = hipEventRecord(r->done[i], s);
    if (e != hipSuccess) return (int)e;
    r->next = (i + 1) % r->depth;
    *slot = i;
    return 0;
}
