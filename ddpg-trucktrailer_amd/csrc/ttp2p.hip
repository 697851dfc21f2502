// ttp2p.hip -- host side of the peer-to-peer gradient exchange (include/ttenv.h: tt_p2p_*; design notes there and in
// DESIGN.md section 5): one block of FINE-GRAINED device memory per rank (arrival words + per-site flat gradient buffers),
// exported as a hipIpcMemHandle and opened by the peers.  Fine-grained because ranks on other GPUs store arrival words into
// it and read gradients from it WHILE kernels of this rank run: coarse-grained memory is only coherent between agents at
// kernel boundaries.  The launch that uses it is k_adam_soft_p2p (csrc/ttlearn.hip).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
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
    static_assert(sizeof(hipIpcMemHandle_t) == TT_P2P_HANDLE_BYTES, "TT_P2P_HANDLE_BYTES");
    static_assert(sizeof(int) * ttp2p::MAXS * ttp2p::MAXR <= ttp2p::HEADER_BYTES, "arrival words fit the header");
    tt_p2p *x = new tt_p2p();
    memset(x, 0, sizeof(*x));
    x->device = device; x->rank = rank; x->world = world; x->sites = sites;
    x->wait_ticks = 200000000ull;      // 2 s of the 100 MHz clock
    size_t off = ttp2p::HEADER_BYTES;
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
    hipError_t e = hipExtMallocWithFlags(&p, x->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return bail(fail(nullptr, TT_ENOMEM, "tt_p2p_create: %zu bytes of fine-grained device memory: %s", x->bytes, hipGetErrorString(e)));
    x->block[rank] = static_cast<char *>(p);
    x->attached[rank] = true;
    if ((e = hipMemset(p, 0, x->bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess)
        return bail(fail(nullptr, TT_EHIP, "tt_p2p_create: clearing the block: %s", hipGetErrorString(e)));
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
        if (r != x->rank && x->attached[r] && x->block[r]) (void)hipIpcCloseMemHandle(x->block[r]);
    if (x->block[x->rank]) (void)hipFree(x->block[x->rank]);
    if (x->gave_up_host) (void)hipHostFree(x->gave_up_host);
    delete x;
    return TT_OK;
}

int tt_p2p_export(const tt_p2p *x, void *handle_out) {
    if (!x || !handle_out) return fail(nullptr, TT_EINVAL, "tt_p2p_export: NULL argument");
    tt_p2p *m = const_cast<tt_p2p *>(x);
    P2P_HIP(m, hipSetDevice(x->device));
    hipIpcMemHandle_t h;
    P2P_HIP(m, hipIpcGetMemHandle(&h, x->block[x->rank]));
    memcpy(handle_out, &h, sizeof(h));
    return TT_OK;
}

int tt_p2p_attach(tt_p2p *x, int peer, const void *handle) {
    if (!x || !handle) return fail(x, TT_EINVAL, "tt_p2p_attach: NULL argument");
    if (peer < 0 || peer >= x->world || peer == x->rank) return fail(x, TT_EINVAL, "tt_p2p_attach: peer %d (this is rank %d of %d)", peer, x->rank, x->world);
    if (x->attached[peer]) return fail(x, TT_EINVAL, "tt_p2p_attach: rank %d is attached already", peer);
    P2P_HIP(x, hipSetDevice(x->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void *p = nullptr;
    P2P_HIP(x, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
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
    P2P_HIP(x, hipMemsetAsync(x->block[x->rank], 0, ttp2p::HEADER_BYTES, stream));
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
