// context.hip -- context lifetime, error string, device buffers, event timers.
#include "common.h"

#include <algorithm>
#include <thread>

static thread_local std::string g_last_error;

void hm_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char* hm_last_error(void) { return g_last_error.c_str(); }
extern "C" int hm_abi_version(void) { return HM_ABI_VERSION; }

extern "C" int hm_create(int device_id, hm_ctx** out) {
    HM_REQUIRE(out != nullptr, "hm_create: out is NULL");
    int ndev = 0;
    HM_HIP(hipGetDeviceCount(&ndev));
    HM_REQUIRE(ndev > 0, "hm_create: no HIP device visible (this library has no CPU fallback)");
    HM_REQUIRE(device_id >= 0 && device_id < ndev, "hm_create: device %d out of range (0..%d)", device_id, ndev - 1);
    HM_HIP(hipSetDevice(device_id));
    hm_ctx* c = new hm_ctx();
    c->device = device_id;
    HM_HIP(hipGetDeviceProperties(&c->prop, device_id));
    c->num_cu = c->prop.multiProcessorCount;
    HM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    return 0;
}

extern "C" int hm_device_count(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return -1;
    return ndev;
}

extern "C" void hm_destroy(hm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int b = 0; b < 2; ++b) {
        if (ctx->pin[b]) (void)hipHostFree(ctx->pin[b]);
        if (ctx->pin_ev[b]) (void)hipEventDestroy(ctx->pin_ev[b]);
    }
    if (ctx->copy_after) (void)hipEventDestroy(ctx->copy_after);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int hm_device_name(hm_ctx* ctx, char* buf, int buflen) {
    HM_REQUIRE(ctx && buf && buflen > 0, "hm_device_name: bad arguments");
    snprintf(buf, buflen, "%s|%s|cu=%d", ctx->prop.gcnArchName, ctx->prop.name, ctx->num_cu);
    return 0;
}

int hm_dev_alloc(DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    HM_HIP(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return 0;
}

void hm_dev_free(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

int EvTimer::begin(hipStream_t s) {
    if (used == evs.size()) {
        hipEvent_t a, b;
        HM_HIP(hipEventCreate(&a));
        HM_HIP(hipEventCreate(&b));
        evs.emplace_back(a, b);
    }
    HM_HIP(hipEventRecord(evs[used].first, s));
    return 0;
}

int EvTimer::end(hipStream_t s) {
    HM_HIP(hipEventRecord(evs[used].second, s));
    ++used;
    return 0;
}

double EvTimer::total_ms() {
    double t = 0;
    for (size_t i = 0; i < used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, evs[i].first, evs[i].second) == hipSuccess) t += ms;
    }
    return t;
}

void EvTimer::destroy() {
    for (auto& e : evs) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    evs.clear();
    used = 0;
}

int hm_d2h_large(hm_ctx* ctx, void* dst_host, const void* src_device, size_t bytes) {
    constexpr size_t PIN = (size_t)64 << 20;
    constexpr int NTHREADS = 8;
    if (bytes < ((size_t)32 << 20)) {
        HM_HIP(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    for (int b = 0; b < 2; ++b) {
        if (!ctx->pin[b]) HM_HIP(hipHostMalloc(&ctx->pin[b], PIN, hipHostMallocDefault));
        if (!ctx->pin_ev[b]) HM_HIP(hipEventCreateWithFlags(&ctx->pin_ev[b], hipEventDisableTiming));
    }
    // pieces of an eighth of the buffer (4 .. 64 MB): a 131 MB ensemble is pipelined as well as a 5 GB history
    const size_t CHUNK = std::min(PIN, std::max((size_t)4 << 20, ((bytes / 8) + 0xFFFFF) & ~(size_t)0xFFFFF));
    const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
    auto issue = [&](size_t k) -> int {
        const size_t off = k * CHUNK, len = std::min(CHUNK, bytes - off);
        HM_HIP(hipMemcpyAsync(ctx->pin[k & 1], (const char*)src_device + off, len, hipMemcpyDeviceToHost, ctx->stream));
        HM_HIP(hipEventRecord(ctx->pin_ev[k & 1], ctx->stream));
        return 0;
    };
    int rc = issue(0);
    if (rc) return rc;
    for (size_t k = 0; k < nchunks; ++k) {
        HM_HIP(hipEventSynchronize(ctx->pin_ev[k & 1]));
        if (k + 1 < nchunks && (rc = issue(k + 1))) return rc;
        const size_t off = k * CHUNK, len = std::min(CHUNK, bytes - off);
        const char* src = (const char*)ctx->pin[k & 1];
        char* dst = (char*)dst_host + off;
        std::thread workers[NTHREADS];
        const size_t slice = ((len / NTHREADS) + 4095) & ~(size_t)4095;
        for (int t = 0; t < NTHREADS; ++t) {
            const size_t a = std::min(len, (size_t)t * slice), e = std::min(len, a + slice);
            workers[t] = std::thread([=]() { if (e > a) memcpy(dst + a, src + a, e - a); });
        }
        for (auto& wkr : workers) wkr.join();
    }
    return 0;
}

// Pageable host -> device copy of a large buffer, the mirror image of hm_d2h_large: worker threads fill one pinned buffer
// while the other goes over PCIe.  Returns when the last piece has been enqueued and the staging buffers are free again;
// ordered on the context's launch stream like the hipMemcpyAsync it replaces.
int hm_h2d_large(hm_ctx* ctx, void* dst_device, const void* src_host, size_t bytes) {
    constexpr size_t PIN = (size_t)64 << 20;
    constexpr int NTHREADS = 8;
    if (bytes < ((size_t)32 << 20)) {
        HM_HIP(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return 0;
    }
    for (int b = 0; b < 2; ++b) {
        if (!ctx->pin[b]) HM_HIP(hipHostMalloc(&ctx->pin[b], PIN, hipHostMallocDefault));
        if (!ctx->pin_ev[b]) HM_HIP(hipEventCreateWithFlags(&ctx->pin_ev[b], hipEventDisableTiming));
    }
    const size_t CHUNK = std::min(PIN, std::max((size_t)4 << 20, ((bytes / 8) + 0xFFFFF) & ~(size_t)0xFFFFF));
    const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
    for (size_t k = 0; k < nchunks; ++k) {
        const size_t off = k * CHUNK, len = std::min(CHUNK, bytes - off);
        if (k >= 2) HM_HIP(hipEventSynchronize(ctx->pin_ev[k & 1]));  // the piece that used this buffer has left it
        char* pin = (char*)ctx->pin[k & 1];
        const char* src = (const char*)src_host + off;
        std::thread workers[NTHREADS];
        const size_t slice = ((len / NTHREADS) + 4095) & ~(size_t)4095;
        for (int t = 0; t < NTHREADS; ++t) {
            const size_t a = std::min(len, (size_t)t * slice), e = std::min(len, a + slice);
            workers[t] = std::thread([=]() { if (e > a) memcpy(pin + a, src + a, e - a); });
        }
        for (auto& wkr : workers) wkr.join();
        HM_HIP(hipMemcpyAsync((char*)dst_device + off, pin, len, hipMemcpyHostToDevice, ctx->stream));
        HM_HIP(hipEventRecord(ctx->pin_ev[k & 1], ctx->stream));
    }
    for (int b = 0; b < 2; ++b) HM_HIP(hipEventSynchronize(ctx->pin_ev[b]));
    return 0;
}

int hm_copy_mark(hm_ctx* ctx) {
    if (!ctx->copy_stream) HM_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->copy_after) HM_HIP(hipEventCreateWithFlags(&ctx->copy_after, hipEventDisableTiming));
    HM_HIP(hipEventRecord(ctx->copy_after, ctx->stream));
    return 0;
}

int hm_d2h_rows(hm_ctx* ctx, void* dst_host, size_t dst_pitch, const void* src_device, size_t src_pitch, size_t width, size_t rows,
                bool marked) {
    constexpr size_t CHUNK = (size_t)64 << 20;
    constexpr int NTHREADS = 8;
    if (!rows || !width) return 0;
    HM_REQUIRE(width <= CHUNK, "hm_d2h_rows: a row of %zu bytes does not fit the staging buffer", width);
    for (int b = 0; b < 2; ++b) {
        if (!ctx->pin[b]) HM_HIP(hipHostMalloc(&ctx->pin[b], CHUNK, hipHostMallocDefault));
        if (!ctx->pin_ev[b]) HM_HIP(hipEventCreateWithFlags(&ctx->pin_ev[b], hipEventDisableTiming));
    }
    int rc = marked ? 0 : hm_copy_mark(ctx);
    if (rc) return rc;
    HM_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_after, 0));
    const size_t per = CHUNK / width, nchunks = (rows + per - 1) / per;
    auto issue = [&](size_t k) -> int {
        const size_t r0 = k * per, nr = std::min(per, rows - r0);
        HM_HIP(hipMemcpy2DAsync(ctx->pin[k & 1], width, (const char*)src_device + r0 * src_pitch, src_pitch, width, nr,
                                hipMemcpyDeviceToHost, ctx->copy_stream));
        HM_HIP(hipEventRecord(ctx->pin_ev[k & 1], ctx->copy_stream));
        return 0;
    };
    if ((rc = issue(0))) return rc;
    for (size_t k = 0; k < nchunks; ++k) {
        HM_HIP(hipEventSynchronize(ctx->pin_ev[k & 1]));
        if (k + 1 < nchunks && (rc = issue(k + 1))) return rc;
        const size_t r0 = k * per, nr = std::min(per, rows - r0);
        const char* src = (const char*)ctx->pin[k & 1];
        char* dst = (char*)dst_host + r0 * dst_pitch;
        std::thread workers[NTHREADS];
        const size_t slice = (nr + NTHREADS - 1) / NTHREADS;
        for (int t = 0; t < NTHREADS; ++t) {
            const size_t a = std::min(nr, (size_t)t * slice), e = std::min(nr, a + slice);
            workers[t] = std::thread([=]() { for (size_t r = a; r < e; ++r) memcpy(dst + r * dst_pitch, src + r * width, width); });
        }
        for (auto& wkr : workers) wkr.join();
    }
    return 0;
}

extern "C" int hm_copy_to_host(hm_ctx* ctx, void* dst_host, const void* src_device, long long bytes) {
    HM_REQUIRE(ctx && dst_host && src_device && bytes >= 0, "hm_copy_to_host: bad arguments");
    HM_HIP(hipSetDevice(ctx->device));
    HM_HIP(hipStreamSynchronize(ctx->stream));
    HM_HIP(hipMemcpy(dst_host, src_device, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int hm_copy_to_device(hm_ctx* ctx, void* dst_device, const void* src_host, long long bytes) {
    HM_REQUIRE(ctx && dst_device && src_host && bytes >= 0, "hm_copy_to_device: bad arguments");
    HM_HIP(hipSetDevice(ctx->device));
    HM_HIP(hipStreamSynchronize(ctx->stream));
    HM_HIP(hipMemcpy(dst_device, src_host, (size_t)bytes, hipMemcpyHostToDevice));
    return 0;
}
