// context.hip -- context lifetime, error string, device buffers, event timers.
#include "common.h"

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

extern "C" void hm_destroy(hm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
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
