// common.h -- context, error handling and small device helpers shared by the HIP sources.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hm_abi.h"

#define HM_ABI_VERSION 2

void hm_set_error(const char* fmt, ...);

#define HM_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            hm_set_error("%s failed at %s:%d: %s", #call, __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define HM_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            hm_set_error(__VA_ARGS__); \
            return 2;                  \
        }                              \
    } while (0)

struct hm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    int num_cu = 0;
    // two pinned staging buffers + events of hm_d2h_large (allocated on first use)
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    // copy stream + event of hm_d2h_rows (results leaving the device while the launch stream computes on)
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_after = nullptr;
};

// Device -> pageable host copy of a large buffer: chunks go to pinned staging buffers at PCIe rate while worker threads
// move the previous chunk into the destination (first-touch page faults included) in parallel.  Synchronous.
int hm_d2h_large(hm_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);
int hm_h2d_large(hm_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);  // the other way; ordered on the launch stream

// The same for `rows` pieces of `width` bytes at a pitch (one time index of every member's history), on the context's COPY
// stream: the copy starts once everything enqueued on the launch stream so far has finished and runs beside whatever is enqueued
// there afterwards.  Synchronous for the host.  "So far" is the moment of the call, or of an earlier hm_copy_mark (marked = true).
int hm_copy_mark(hm_ctx* ctx);
int hm_d2h_rows(hm_ctx* ctx, void* dst_host, size_t dst_pitch, const void* src_device, size_t src_pitch, size_t width, size_t rows,
                bool marked = false);

// RAII-less device buffer bookkeeping (plans free what they allocate).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

int hm_dev_alloc(DevBuf& b, size_t bytes);
void hm_dev_free(DevBuf& b);

// Event pair timing accumulated per kernel class.
struct EvTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    size_t used = 0;
    int begin(hipStream_t s);
    int end(hipStream_t s);
    double total_ms();  // requires stream synchronised
    void reset() { used = 0; }
    void destroy();
};
