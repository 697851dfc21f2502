// ttp2p.hip -- host side of the peer-to-peer gradient exchange (include/ttenv.h: tt_p2p_*; design notes there and in
// DESIGN.md section 5): two allocations per rank, each exported as a hipIpcMemHandle and opened by the peers --
//   a FLAG block of fine-grained device memory (the arrival words): ranks on other GPUs store into it WHILE kernels of this
//     rank poll it, and coarse-grained memory is only coherent between agents at kernel boundaries;
//   a GRADIENT block of ordinary (coarse-grained) device memory, per site a flat f32 buffer: written by this rank's backward
//     launch like any other buffer (L2-cached, written back when that launch ends), read by the peers AFTER the arrival word
//     -- which is stored by a later launch of the same stream -- with system-scope loads that no cache of theirs serves.
// The launch that uses both is k_adam_soft_p2p (csrc/ttlearn.hip).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ttp2p.h"

namespace {

char g_err[256] = "";

int fail(tt_p2p *x, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(x ? x->err : g_err, 256, fmt, ap);
    va_end(ap);
    return code;
}

#define P2P_HIP(x, call)                                                                                   \
    do {                                                                                                   \
        const hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(x, TT_EHIP, "%s: %s", #call, hipGetErrorString(e_));             \
    } while (0)

}  // namespace

extern "C" {

const char *tt_p2p_last_error(const tt_p2p *x) { return x ? x->err : g_err; }

int tt_p2p_create(int device, int rank, int world, int sites, const int32_t *numel, tt_p2p **out) {
    if (!out || !numel || world < 1 || world > ttp2p::MAXR || rank < 0 || rank >= world || sites < 1 || sites > ttp2p::MAXS)
        return fail(nullptr, TT_EINVAL, "tt_p2p_create: rank %d of %d (at most %d), %d sites (at most %d)", rank, world, ttp2p::MAXR,
                    sites, ttp2p::MAXS);
    static_assert(2 * sizeof(hipIpcMemHandle_t) == TT_P2P_HANDLE_BYTES, "TT_P2P_HANDLE_BYTES");
    static_assert(sizeof(int) * ttp2p::MAXS * ttp2p::MAXR <= ttp2p::FLAG_BYTES, "arrival words fit the flag block");
    tt_p2p *x = new tt_p2p();
    memset(x, 0, sizeof(*x));
    x->device = device; x->rank = rank; x->world = world; x->sites = sites;
    x->wait_ticks = 200000000ull;      // 2 s of the 100 MHz clock
    size_t off = 0;
    for (int s = 0; s < sites; ++s) {
        if (numel[s] <= 0) { delete x; return fail(nullptr, TT_EINVAL, "tt_p2p_create: numel[%d] = %d", s, numel[s]); }
        x->numel[s] = numel[s];
        x->offset[s] = off;
        off += ((size_t)numel[s] * sizeof(float) + 255) / 256 * 256;
    }
    x->bytes = off;
    const auto bail = [&](int code) { tt_p2p_destroy(x); return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(nullptr, TT_ENODEV, "tt_p2p_create: no HIP device %d", device));
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, ttp2p::FLAG_BYTES, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return bail(fail(nullptr, TT_ENOMEM, "tt_p2p_create: fine-grained device memory for the arrival words: %s", hipGetErrorString(e)));
    x->flags[rank] = static_cast<char *>(p);
    const char *fg = getenv("TT_P2P_FINE_GRADS");
    x->fine_grads = fg && fg[0] == '1';
    p = nullptr;
    e = x->fine_grads ? hipExtMallocWithFlags(&p, x->bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, x->bytes);
    if (e != hipSuccess) return bail(fail(nullptr, TT_ENOMEM, "tt_p2p_create: %zu bytes of device memory for the gradient buffers: %s", x->bytes, hipGetErrorString(e)));
    x->block[rank] = static_cast<char *>(p);
    x->attached[rank] = true;
    if ((e = hipMemset(x->flags[rank], 0, ttp2p::FLAG_BYTES)) != hipSuccess || (e = hipMemset(p, 0, x->bytes)) != hipSuccess ||
        (e = hipDeviceSynchronize()) != hipSuccess)
        return bail(fail(nullptr, TT_EHIP, "tt_p2p_create: clearing the blocks: %s", hipGetErrorString(e)));
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&x->gave_up_host), 64, hipHostMallocDefault)) != hipSuccess)
        return bail(fail(nullptr, TT_ENOMEM, "tt_p2p_create: pinned host word: %s", hipGetErrorString(e)));
    *x->gave_up_host = 0;
    *out = x;
    return TT_OK;
}

int tt_p2p_destroy(tt_p2p *x) {
    if (!x) return TT_OK;
    (void)hipSetDevice(x->device);
    for (int r = 0; r < x->world; ++r)
        if (r != x->rank) {
            if (x->block[r]) (void)hipIpcCloseMemHandle(x->block[r]);
            if (x->flags[r]) (void)hipIpcCloseMemHandle(x->flags[r]);
        }
    if (x->block[x->rank]) (void)hipFree(x->block[x->rank]);
    if (x->flags[x->rank]) (void)hipFree(x->flags[x->rank]);
    if (x->gave_up_host) (void)hipHostFree(x->gave_up_host);
    delete x;
    return TT_OK;
}

int tt_p2p_export(const tt_p2p *x, void *handle_out) {
    if (!x || !handle_out) return fail(nullptr, TT_EINVAL, "tt_p2p_export: NULL argument");
    tt_p2p *m = const_cast<tt_p2p *>(x);
    P2P_HIP(m, hipSetDevice(x->device));
    hipIpcMemHandle_t h[2];
    P2P_HIP(m, hipIpcGetMemHandle(&h[0], x->flags[x->rank]));
    P2P_HIP(m, hipIpcGetMemHandle(&h[1], x->block[x->rank]));
    memcpy(handle_out, h, sizeof(h));
    return TT_OK;
}

int tt_p2p_attach(tt_p2p *x, int peer, const void *handle) {
    if (!x || !handle) return fail(x, TT_EINVAL, "tt_p2p_attach: NULL argument");
    if (peer < 0 || peer >= x->world || peer == x->rank) return fail(x, TT_EINVAL, "tt_p2p_attach: peer %d (this is rank %d of %d)", peer, x->rank, x->world);
    if (x->attached[peer]) return fail(x, TT_EINVAL, "tt_p2p_attach: rank %d is attached already", peer);
    P2P_HIP(x, hipSetDevice(x->device));
    hipIpcMemHandle_t h[2];
    memcpy(h, handle, sizeof(h));
    void *p = nullptr;
    P2P_HIP(x, hipIpcOpenMemHandle(&p, h[0], hipIpcMemLazyEnablePeerAccess));
    x->flags[peer] = static_cast<char *>(p);
    p = nullptr;
    P2P_HIP(x, hipIpcOpenMemHandle(&p, h[1], hipIpcMemLazyEnablePeerAccess));
    x->block[peer] = static_cast<char *>(p);
    x->attached[peer] = true;
    return TT_OK;
}

float *tt_p2p_grad(const tt_p2p *x, int site) {
    if (!x || site < 0 || site >= x->sites) return nullptr;
    return reinterpret_cast<float *>(x->block[x->rank] + x->offset[site]);
}

int tt_p2p_reset(tt_p2p *x, tt_stream_t stream) {
    if (!x) return fail(nullptr, TT_EINVAL, "tt_p2p_reset: NULL handle");
    P2P_HIP(x, hipSetDevice(x->device));
    P2P_HIP(x, hipMemsetAsync(x->flags[x->rank], 0, ttp2p::FLAG_BYTES, stream));
    P2P_HIP(x, hipStreamSynchronize(stream));
    *x->gave_up_host = 0;
    return TT_OK;
}

int tt_p2p_set_timeout(tt_p2p *x, double seconds) {
    if (!x || !(seconds > 0.0) || seconds > 3600.0) return fail(x, TT_EINVAL, "tt_p2p_set_timeout: %g s", seconds);
    x->wait_ticks = (unsigned long long)(seconds * 1e8);
    return TT_OK;
}

int tt_p2p_gave_up(const tt_p2p *x) { return x ? *reinterpret_cast<volatile int *>(x->gave_up_host) : 0; }

}  // extern "C"
