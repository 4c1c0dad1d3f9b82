// api.hip -- context, error plumbing, memory helpers and timers of the C ABI (include/wsscam.h).
#include "common.h"

#include <cerrno>
#include <cstdarg>
#include <fcntl.h>
#include <sys/uio.h>
#include <unistd.h>
#include <vector>
#include <cstdio>
#include <cstring>

static thread_local char g_err[1024] = "";

void wsc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int wsc_ctx_workspace(wsc_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->ws_bytes) {
        // everything already enqueued may still read the old arena
        WSC_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->ws) WSC_HIP(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
        const size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&ctx->ws, want);
        if (e != hipSuccess) {
            wsc_set_error("workspace hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            return WSC_ERR_NOMEM;
        }
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return WSC_OK;
}

int wsc_ctx_cached_alloc(wsc_ctx *ctx, size_t bytes, void **out) {
    size_t want = bytes < 256 ? 256 : bytes;
    want = want <= (1u << 20) ? (want + 4095) / 4096 * 4096 : (want + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    auto it = ctx->free_blocks.lower_bound(want);
    if (it != ctx->free_blocks.end() && it->first <= 2 * want + (1u << 20)) {
        *out = it->second;
        ctx->live_blocks[it->second] = it->first;
        ctx->free_blocks.erase(it);
        return WSC_OK;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        // drop the cache and retry once
        (void)hipStreamSynchronize(ctx->stream);
        for (auto &kv : ctx->free_blocks) (void)hipFree(kv.second);
        ctx->free_blocks.clear();
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        wsc_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return WSC_ERR_NOMEM;
    }
    ctx->live_blocks[p] = want;
    *out = p;
    return WSC_OK;
}

void wsc_ctx_cached_free(wsc_ctx *ctx, void *p) {
    if (!p) return;
    auto it = ctx->live_blocks.find(p);
    if (it == ctx->live_blocks.end()) return;
    ctx->free_blocks.emplace(it->second, p);
    ctx->live_blocks.erase(it);
}

int wsc_ctx_upload_small(wsc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (bytes == 0) return WSC_OK;
    wsc_ctx::PinSlot &s = ctx->pin_ring[ctx->pin_next];
    ctx->pin_next = (ctx->pin_next + 1) % wsc_ctx::PIN_SLOTS;
    if (s.busy) { // the ring has gone round: this slot's previous copy must have left it
        WSC_HIP(hipEventSynchronize(s.ev));
        s.busy = false;
    }
    if (!s.ev) WSC_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    if (bytes > s.bytes) {
        if (s.p) WSC_HIP(hipHostFree(s.p));
        s.p = nullptr;
        s.bytes = 0;
        const size_t want = bytes * 2 < (64u << 10) ? (64u << 10) : bytes * 2;
        WSC_HIP(hipHostMalloc(&s.p, want, hipHostMallocDefault));
        s.bytes = want;
    }
    memcpy(s.p, src_host, bytes);
    WSC_HIP(hipMemcpyAsync(dst_dev, s.p, bytes, hipMemcpyHostToDevice, ctx->stream));
    WSC_HIP(hipEventRecord(s.ev, ctx->stream));
    s.busy = true;
    return WSC_OK;
}

static hipEvent_t prof_event(wsc_ctx *ctx) {
    if (!ctx->prof_pool.empty()) {
        hipEvent_t e = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

WscKernelTimer::WscKernelTimer(wsc_ctx *c, int cls, double work) : ctx(c) {
    if (!ctx->profiling) return;
    WscProfRecord r;
    r.cls = cls;
    r.work = work;
    r.e0 = prof_event(ctx);
    r.e1 = prof_event(ctx);
    (void)hipEventRecord(r.e0, ctx->stream);
    ctx->prof.push_back(r);
    idx = (int)ctx->prof.size() - 1;
}
WscKernelTimer::~WscKernelTimer() {
    if (idx >= 0) (void)hipEventRecord(ctx->prof[idx].e1, ctx->stream);
}

static const char *kClassNames[WSC_K_COUNT] = {
    "conv_igemm_kernel<256-row tiles,glds>", "conv_igemm_kernel<128x128,glds>", "conv_igemm_kernel<128x64,glds>", "conv_igemm_kernel<small-Cin> / stem_pool_kernel",
    "pool/layout/flip-add", "cam_tail+unary", "crf_build(all)", "gauss_msg_kernel", "combine4+blur_lds+blur4+blur3_tile",
    "update_splat_kernel", "crf init/finish"};

int wsc_ctx_range_check(wsc_ctx *ctx) {
    const unsigned f = *(volatile unsigned *)ctx->range_host;
    if (f == 0u) return WSC_OK;
    wsc_set_error("an activation of an IEEE-half conv mode (f16 / f16x3) reached half's ceiling (|v| >= 65504) in a layer with %u "
                  "output channels and was saturated -- the reference's fp32 would have kept it, these maps are not the reference's; "
                  "use WSC_PREC_BF16X3 for this model (wsc_ctx_range_status clears the flag)", f);
    return WSC_ERR_RANGE;
}

extern "C" {

int wsc_profile_begin(wsc_ctx *ctx) {
    WSC_CHECK(ctx, WSC_ERR_INVALID, "wsc_profile_begin: null ctx");
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof) {
        ctx->prof_pool.push_back(r.e0);
        ctx->prof_pool.push_back(r.e1);
    }
    ctx->prof.clear();
    ctx->profiling = true;
    return WSC_OK;
}

int wsc_profile_end(wsc_ctx *ctx, int max_classes, int32_t *calls_out, float *total_ms_out, double *work_out,
                    int *n_classes_out) {
    WSC_CHECK(ctx && calls_out && total_ms_out && work_out, WSC_ERR_INVALID, "wsc_profile_end: null argument");
    ctx->profiling = false;
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    const int n = max_classes < WSC_K_COUNT ? max_classes : (int)WSC_K_COUNT;
    for (int i = 0; i < n; ++i) {
        calls_out[i] = 0;
        total_ms_out[i] = 0.f;
        work_out[i] = 0.0;
    }
    for (auto &r : ctx->prof) {
        float ms = 0.f;
        WSC_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        if (r.cls < n) {
            calls_out[r.cls] += 1;
            total_ms_out[r.cls] += ms;
            work_out[r.cls] += r.work;
        }
        ctx->prof_pool.push_back(r.e0);
        ctx->prof_pool.push_back(r.e1);
    }
    ctx->prof.clear();
    if (n_classes_out) *n_classes_out = n;
    return WSC_OK;
}

const char *wsc_profile_class_name(int cls) { return cls >= 0 && cls < WSC_K_COUNT ? kClassNames[cls] : ""; }

int wsc_version(void) { return WSC_VERSION; }
const char *wsc_last_error(void) { return g_err; }

int wsc_ctx_create(int device, void *stream, wsc_ctx **out) {
    WSC_CHECK(out != nullptr, WSC_ERR_INVALID, "wsc_ctx_create: out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        wsc_set_error("no HIP device available (%s); libwsscam has no CPU fallback",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return WSC_ERR_NO_DEVICE;
    }
    WSC_CHECK(device >= 0 && device < count, WSC_ERR_INVALID, "device %d out of range [0,%d)", device, count);
    WSC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    WSC_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        wsc_set_error("device %d is %s; libwsscam is built for gfx950 only", device, prop.gcnArchName);
        return WSC_ERR_NO_DEVICE;
    }
    wsc_ctx *ctx = new wsc_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    ctx->arch = prop.gcnArchName;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        WSC_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    WSC_HIP(hipMalloc(&ctx->zero_page, 256));
    WSC_HIP(hipMemset(ctx->zero_page, 0, 256));
    WSC_HIP(hipHostMalloc((void **)&ctx->range_host, 64, hipHostMallocMapped));
    memset(ctx->range_host, 0, 64);
    WSC_HIP(hipHostGetDevicePointer((void **)&ctx->range_dev, ctx->range_host, 0));
    WSC_HIP(hipEventCreateWithFlags(&ctx->pinned_ev, hipEventDisableTiming));
    WSC_HIP(hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming));
    WSC_HIP(hipEventCreate(&ctx->ev0));
    WSC_HIP(hipEventCreate(&ctx->ev1));
    *out = ctx;
    return WSC_OK;
}

int wsc_ctx_set_option(wsc_ctx *ctx, int option, int value) {
    WSC_CHECK(ctx != nullptr, WSC_ERR_INVALID, "wsc_ctx_set_option: null context");
    WSC_CHECK(option >= 0 && option < WSC_OPT_COUNT, WSC_ERR_INVALID, "wsc_ctx_set_option: unknown option %d", option);
    ctx->opt[option] = value;
    return WSC_OK;
}

void wsc_ctx_destroy(wsc_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &a : ctx->attachments) a.second(a.first);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->zero_page) (void)hipFree(ctx->zero_page);
    if (ctx->range_host) (void)hipHostFree(ctx->range_host);
    for (auto &kv : ctx->free_blocks) (void)hipFree(kv.second);
    for (auto &kv : ctx->live_blocks) (void)hipFree(kv.first);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (wsc_ctx::PinSlot &s : ctx->pin_ring) {
        if (s.p) (void)hipHostFree(s.p);
        if (s.ev) (void)hipEventDestroy(s.ev);
    }
    if (ctx->pinned_ev) (void)hipEventDestroy(ctx->pinned_ev);
    if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
    for (hipEvent_t e : ctx->marks)
        if (e) (void)hipEventDestroy(e);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->aux_done_ev) (void)hipEventDestroy(ctx->aux_done_ev);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
    }
    for (auto &r : ctx->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int wsc_sync(wsc_ctx *ctx) {
    WSC_CHECK(ctx, WSC_ERR_INVALID, "wsc_sync: null ctx");
    WSC_HIP(hipSetDevice(ctx->device)); // may be called from a helper thread whose current device is still 0
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    return wsc_ctx_range_check(ctx);
}

int wsc_ctx_range_status(wsc_ctx *ctx, int *flag_out, int clear) {
    WSC_CHECK(ctx, WSC_ERR_INVALID, "wsc_ctx_range_status: null ctx");
    WSC_HIP(hipSetDevice(ctx->device));
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    volatile unsigned *f = ctx->range_host;
    if (flag_out) *flag_out = (int)*f;
    if (clear) *f = 0u;
    return WSC_OK;
}

int wsc_ctx_mark(wsc_ctx *ctx, int slot) {
    WSC_CHECK(ctx && slot >= 0 && slot < 8, WSC_ERR_INVALID, "wsc_ctx_mark: slot %d outside [0, 8)", slot);
    WSC_HIP(hipSetDevice(ctx->device));
    if (!ctx->marks[slot]) WSC_HIP(hipEventCreateWithFlags(&ctx->marks[slot], hipEventDisableTiming));
    WSC_HIP(hipEventRecord(ctx->marks[slot], ctx->stream));
    ctx->mark_set[slot] = true;
    return WSC_OK;
}

int wsc_ctx_wait_mark(wsc_ctx *ctx, int slot) {
    WSC_CHECK(ctx && slot >= 0 && slot < 8, WSC_ERR_INVALID, "wsc_ctx_wait_mark: slot %d outside [0, 8)", slot);
    if (!ctx->mark_set[slot]) return WSC_OK;
    WSC_HIP(hipSetDevice(ctx->device));
    WSC_HIP(hipEventSynchronize(ctx->marks[slot]));
    return wsc_ctx_range_check(ctx);
}

int wsc_ctx_wait_for_mark(wsc_ctx *ctx, wsc_ctx *other, int slot) {
    WSC_CHECK(ctx && other && slot >= 0 && slot < 8, WSC_ERR_INVALID, "wsc_ctx_wait_for_mark: bad argument (slot %d)", slot);
    WSC_CHECK(ctx->device == other->device, WSC_ERR_INVALID, "wsc_ctx_wait_for_mark: contexts are on different devices");
    if (!other->mark_set[slot] || ctx == other) return WSC_OK;
    WSC_HIP(hipSetDevice(ctx->device));
    WSC_HIP(hipStreamWaitEvent(ctx->stream, other->marks[slot], 0));
    return WSC_OK;
}

int wsc_ctx_wait(wsc_ctx *ctx, wsc_ctx *other) {
    WSC_CHECK(ctx && other, WSC_ERR_INVALID, "wsc_ctx_wait: null ctx");
    WSC_CHECK(ctx->device == other->device, WSC_ERR_INVALID, "wsc_ctx_wait: contexts are on different devices");
    if (ctx == other) return WSC_OK;
    WSC_HIP(hipEventRecord(other->join_ev, other->stream));
    WSC_HIP(hipStreamWaitEvent(ctx->stream, other->join_ev, 0));
    return WSC_OK;
}

int wsc_device_info(wsc_ctx *ctx, char *arch_name, size_t arch_name_len, int *num_cus) {
    WSC_CHECK(ctx, WSC_ERR_INVALID, "wsc_device_info: null ctx");
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, ctx->arch.c_str(), arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    if (num_cus) *num_cus = ctx->num_cus;
    return WSC_OK;
}

int wsc_malloc(wsc_ctx *ctx, size_t bytes, void **dptr_out) {
    WSC_CHECK(ctx && dptr_out, WSC_ERR_INVALID, "wsc_malloc: null argument");
    WSC_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr_out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        wsc_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return WSC_ERR_NOMEM;
    }
    return WSC_OK;
}

int wsc_free(wsc_ctx *ctx, void *dptr) {
    WSC_CHECK(ctx, WSC_ERR_INVALID, "wsc_free: null ctx");
    if (!dptr) return WSC_OK;
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    WSC_HIP(hipFree(dptr));
    return WSC_OK;
}

int wsc_memcpy_h2d(wsc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    WSC_CHECK(ctx && (bytes == 0 || (dst_dev && src_host)), WSC_ERR_INVALID, "wsc_memcpy_h2d: null argument");
    if (bytes == 0) return WSC_OK;
    WSC_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable source: the runtime has staged it when the call returns
    return WSC_OK;
}

int wsc_memcpy_d2h(wsc_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    WSC_CHECK(ctx && (bytes == 0 || (dst_host && src_dev)), WSC_ERR_INVALID, "wsc_memcpy_d2h: null argument");
    if (bytes == 0) return WSC_OK;
    // Wait for the producers on the host FIRST, then copy: a copy queued behind unfinished kernels becomes a poll command on the
    // DMA engine's in-order queue and holds back every copy submitted after it, other contexts' uploads included, for as long as
    // those kernels run (round 6, profiles/r06_step_timeline_e2e_before.txt)
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    WSC_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    return wsc_ctx_range_check(ctx);
}

int wsc_memset(wsc_ctx *ctx, void *dst_dev, int value, size_t bytes) {
    WSC_CHECK(ctx && (bytes == 0 || dst_dev), WSC_ERR_INVALID, "wsc_memset: null argument");
    if (bytes == 0) return WSC_OK;
    WSC_HIP(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return WSC_OK;
}

int wsc_host_alloc(wsc_ctx *ctx, size_t bytes, void **host_out) {
    WSC_CHECK(ctx && host_out, WSC_ERR_INVALID, "wsc_host_alloc: null argument");
    WSC_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(host_out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        wsc_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return WSC_ERR_NOMEM;
    }
    return WSC_OK;
}

int wsc_host_free(wsc_ctx *ctx, void *host) {
    WSC_CHECK(ctx, WSC_ERR_INVALID, "wsc_host_free: null ctx");
    if (!host) return WSC_OK;
    WSC_HIP(hipStreamSynchronize(ctx->stream));
    WSC_HIP(hipHostFree(host));
    return WSC_OK;
}

int wsc_host_write_segments(const char *path, int n, const void *const *ptrs, const size_t *sizes) {
    WSC_CHECK(path && n >= 0 && (n == 0 || (ptrs && sizes)), WSC_ERR_INVALID, "wsc_host_write_segments: null argument");
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        wsc_set_error("wsc_host_write_segments: cannot open %s: %s", path, strerror(errno));
        return WSC_ERR_INVALID;
    }
    int st = WSC_OK;
    std::vector<struct iovec> iov;
    for (int i = 0; i < n; ++i)
        if (sizes[i] > 0) iov.push_back({const_cast<void *>(ptrs[i]), sizes[i]});
    size_t at = 0;
    while (at < iov.size() && st == WSC_OK) {
        const int cnt = (int)std::min<size_t>(iov.size() - at, 64);
        ssize_t w = writev(fd, iov.data() + at, cnt);
        if (w < 0) {
            if (errno == EINTR) continue;
            wsc_set_error("wsc_host_write_segments: write to %s failed: %s", path, strerror(errno));
            st = WSC_ERR_INVALID;
            break;
        }
        while (w > 0 && at < iov.size()) { // advance past what was written (short writes leave a partial segment)
            if ((size_t)w >= iov[at].iov_len) {
                w -= (ssize_t)iov[at].iov_len;
                ++at;
            } else {
                iov[at].iov_base = (char *)iov[at].iov_base + w;
                iov[at].iov_len -= (size_t)w;
                w = 0;
            }
        }
    }
    if (close(fd) != 0 && st == WSC_OK) {
        wsc_set_error("wsc_host_write_segments: close of %s failed: %s", path, strerror(errno));
        st = WSC_ERR_INVALID;
    }
    return st;
}

int wsc_memcpy_h2d_async(wsc_ctx *ctx, void *dst_dev, const void *src_pinned_host, size_t bytes) {
    WSC_CHECK(ctx && (bytes == 0 || (dst_dev && src_pinned_host)), WSC_ERR_INVALID, "wsc_memcpy_h2d_async: null argument");
    if (bytes == 0) return WSC_OK;
    WSC_HIP(hipMemcpyAsync(dst_dev, src_pinned_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return WSC_OK;
}

int wsc_memcpy_d2h_async(wsc_ctx *ctx, void *dst_pinned_host, const void *src_dev, size_t bytes) {
    WSC_CHECK(ctx && (bytes == 0 || (dst_pinned_host && src_dev)), WSC_ERR_INVALID, "wsc_memcpy_d2h_async: null argument");
    if (bytes == 0) return WSC_OK;
    WSC_HIP(hipMemcpyAsync(dst_pinned_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return WSC_OK;
}

int wsc_timer_begin(wsc_ctx *ctx) {
    WSC_CHECK(ctx, WSC_ERR_INVALID, "wsc_timer_begin: null ctx");
    WSC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return WSC_OK;
}

int wsc_timer_end(wsc_ctx *ctx, float *ms_out) {
    WSC_CHECK(ctx && ms_out, WSC_ERR_INVALID, "wsc_timer_end: null argument");
    WSC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    WSC_HIP(hipEventSynchronize(ctx->ev1));
    WSC_HIP(hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
    return WSC_OK;
}

} // extern "C"
