// ttlearn.hip -- hand-fused DDPG learn() for the reference-shaped networks (23 -> 400 -> LN -> ReLU -> 300 -> LN ...),
// exact f32, MI355X (gfx950).  What DDPG_agent.learn() (DDPG/DDPG_agent.py:72-106) does through ~140 autograd
// kernels at batch 256 is done here in six launches (sample -- csrc/ttnet.hip -- then):
//
//   k_fwd_multi           learn()'s first phase: up to four forwards in one launch (target actor and the target critic's
//                         state branch on s', Q(s,a) and mu(s) with what their backward needs: normalised
//                         pre-activations, 1/sigma, post-ReLU activations); k_fwd_small<CRITIC> is one of them alone
//   k_bwd_rows<CRITIC>    per-row backward: [critic: q' and the TD target for its rows ->] head -> ReLU -> LayerNorm2 ->
//                         dH1 = dX2 * W2 (MFMA) -> ReLU -> LayerNorm1
//   k_bwd_weights         dW2 = dX2^T * H1 and dW1 = dX1^T * S on the MFMA (K = batch), all bias / LayerNorm / head
//                         gradients as deterministic column sums (no atomics), then (one rank) torch.optim.Adam's
//                         update + the soft target update on each element just finished
//   k_adam_soft           Adam + soft update as a launch of its own (data-parallel ranks: after the all-reduce);
//   k_head_td, k_td_target  the TD target as launches of their own
//
// Small-batch geometry: a workgroup owns 16 rows; its 8 waves split the output COLUMNS (so a 256-row batch is 16
// workgroups x 8 waves), and LayerNorm statistics are combined across the waves through LDS.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ttenv.h"
#include "ttnet_common.h"      // the replay draw (ring_sample_index): k_fwd_multi can make it itself
#include "ttnet_pack.h"        // the policy image (split_pack_body): k_bwd_rows_pair can carry its pack
#include "ttp2p.h"
#include "ttstamps.h"             // the peer-to-peer gradient exchange of data-parallel ranks: k_adam_soft_p2p reads it

namespace {

constexpr int IN = 23, INP = 24;
constexpr int H1 = 400, H2 = 300, H2P = 320, H2K = 304;   // H2K: fc2 outputs rounded up to whole k16 steps
constexpr int NT1 = H1 / 16, NT2 = H2P / 16;
constexpr int TR = 16;                    // rows per workgroup
constexpr int NW = 8;                     // waves per workgroup: two per SIMD, so one wave's load latency hides behind
                                          // the other's MFMAs (hipcc does not keep a deep software prefetch in place)
constexpr int HS1 = 404;                  // LDS row stride of the 16 x 400 activation tile: 16-byte rows, and 404 mod 64 = 20
                                          // spreads the 16 rows of a ds_read_b128 fragment over all 64 banks

constexpr int DS = 308;                  // LDS row stride of the 16 x 300 tiles (fc2 pre-activations; dX2 as the A operand
                                          // with K = 304): 16-byte rows

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// ---- fc2 products on the f16 MFMA from a pre-split image of the weights (tt_mlp_weights.fc2_img; include/ttenv.h) ----
// An f32 number is, to within its own rounding, the sum of two round-to-nearest f16 numbers (csrc/ttnet_split.hip has the
// argument): x*w = h*h' + h*m' + m*h' to ~2^-22, three v_mfma_f32_16x16x32_f16 per block at 16x the f32 MFMA rate.  The
// weights' pieces come ready from the image (the optimizer launches keep it current element by element), the activations
// / gradients of a workgroup's 16 rows are split once into LDS planes.  Scales: activations x16, weights x64, gradient rows
// by a per-row power of two (so the m pieces stay normal); all exact.
constexpr int K2P = 416;                 // fc2 inputs padded to whole k32 steps (13)
constexpr int N2P = 320;                 // fc2 outputs padded to whole k32 steps (10) / tiles (20)
constexpr int HSH = 424;                 // LDS row stride (halves) of the 16 x 416 planes: 212 dwords, 212 mod 64 = 20 (as HS1)
constexpr int DSH = 328;                 // ... of the 16 x 320 planes: 164 dwords, 164 mod 64 = 36: a b128 fragment read is conflict-free
// The image holds MFMA fragments, 1 KB each (64 lanes x 8 halves = what ONE global_load_dwordx4 of a wave fetches, fully
// coalesced), in two orientations, an h and an m plane of each:
//   forward  [20 tiles][13 k32 steps][64 lanes][8]: lane (l15, l4) of tile T holds W2[n = 16 T + l15][k = 32 s + 8 l4 + j];
//   backward [28 tiles][10 k32 steps][64 lanes][8]: dH1 = dX2 * W2 sums over n; tile T = 4 g + t serves output column
//            k = 64 g + 4 l15 + t (the accumulator mapping of the f32 path: a lane's four tiles are four consecutive columns),
//            lane (l15, l4) holds W2[n = 32 s + 8 l4 + j][k]; tiles beyond column 399 stay zero.
constexpr int FW_TILES = N2P / 16, FW_STEPS = K2P / 32, BW_TILES = 28, BW_STEPS = N2P / 32;
constexpr size_t IMG_FWD = (size_t)FW_TILES * FW_STEPS * 512, IMG_T = (size_t)BW_TILES * BW_STEPS * 512;   // halves per plane
constexpr size_t IMG_HALVES = 2 * IMG_FWD + 2 * IMG_T;                       // 1,105,920 bytes
constexpr float SXL = 16.f, SWL = 64.f, UNSC_L = 1.f / (SXL * SWL);
constexpr int H1S_FLOATS = 16 * HSH;     // floats of the activation tile's LDS buffer: 16 x 404 f32, or the two 16 x 424 f16 planes
constexpr int DXS_FLOATS = 16 * DSH;     // ... of the dX2 tile's: 16 x 308 f32, or the two 16 x 328 f16 planes
__device__ __forceinline__ f32x4 mfma_h(const f16x8 a, const f16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// (stores of the optimizer state / updated weights / image patches: plain.  Non-temporal ones -- nothing of the same launch reads
// them again -- made learn() 2 us SLOWER, 66.9 -> 68.9 us: the next launch's row kernels then fetch the weights from memory)
__device__ __forceinline__ void st_out(float *p, const float v) { *p = v; }
__device__ __forceinline__ void st_out16(_Float16 *p, const uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }
// the two pieces of one fc2 weight into an image: element (n = output neuron, k = input)
__device__ __forceinline__ void img_store(_Float16 *__restrict__ img, const int n, const int k, const float w, const bool with_t) {
    const float s = w * SWL;
    const _Float16 h = (_Float16)s, m = (_Float16)(s - (float)h);
    const size_t f = ((size_t)((n >> 4) * FW_STEPS + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7);
    img[f] = h;
    img[IMG_FWD + f] = m;
    if (with_t) {
        const int tile = (k >> 6) * 4 + (k & 3), l15 = (k >> 2) & 15;
        const size_t b = ((size_t)(tile * BW_STEPS + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + l15) * 8 + (n & 7);
        img[2 * IMG_FWD + b] = h;
        img[2 * IMG_FWD + IMG_T + b] = m;
    }
}

#ifdef TT_STAMPS   // diagnostic build only: wall-clock stamps (100 MHz) of workgroup 0 / wave 0 at phase boundaries
__device__ unsigned long long g_stamps[32];
__device__ unsigned long long g_blk[512][2];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_stamps[i] = wall_clock64(); g_stamps[16 + i] = clock64(); } } while (0)
#define STAMPB(i, blk0) do { if ((int)blockIdx.x == (blk0) && threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)
__device__ unsigned long long g_sub[16];        // finer stamps inside the row forward's phases (workgroup 0; tools/learn_blocks.py)
#define SUB(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sub[i] = wall_clock64(); } while (0)
__device__ unsigned long long g_wst[2][16];     // phase stamps of workgroup 0 of k_bwd_weights<critic / actor>
#define WST(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_wst[ROWSCALE ? 1 : 0][i] = wall_clock64(); } while (0)
// begin / end of every workgroup of learn()'s launches: [kernel][block][2] (tools/learn_blocks.py)
__device__ unsigned long long g_kblk[6][512][2];
#define KBEGIN(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_kblk[k][blockIdx.x][0] = wall_clock64(); } while (0)
__device__ TTLog g_log_learn[5];
#define KEND(k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 512) { g_kblk[k][blockIdx.x][1] = wall_clock64();       \
        if ((k) < 5) tt_log_add(g_log_learn[(k) < 5 ? (k) : 0], g_kblk[k][blockIdx.x][0], g_kblk[k][blockIdx.x][1]); } } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMPB(i, blk0) do { } while (0)
#define SUB(i) do { } while (0)
#define KBEGIN(k) do { } while (0)
#define KEND(k) do { } while (0)
#define WST(i) do { } while (0)
#endif

// Kernel arguments of these launches are structs of a few hundred bytes to over a kilobyte (k_fwd_multi: four jobs of thirteen
// weight pointers each + the replay draw).  The compiler reads them from the kernarg segment lazily, field group by field group,
// each group with its own `s_waitcnt lgkmcnt(0)` -- and every first touch of a 64-byte line misses the scalar cache: a row kernel
// had three to five such round trips spread over its critical path.  kernarg_warm<BYTES>() touches every line of the first BYTES of
// the segment at the top of the kernel, all loads in flight together, one wait: the later reads hit.
template <int OFF>
__device__ __forceinline__ unsigned kernarg_touch_line(const void *ka) {
    unsigned r;
    asm volatile("s_load_dword %0, %1, %2" : "=s"(r) : "s"(ka), "i"(OFF));
    return r;
}
template <int LINE, int LINES>
__device__ __forceinline__ void kernarg_touch_all(const void *ka, unsigned (&t)[LINES]) {
    if constexpr (LINE < LINES) {
        t[LINE] = kernarg_touch_line<LINE * 64>(ka);
        kernarg_touch_all<LINE + 1, LINES>(ka, t);
    }
}
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#ifndef TT_DBG_NO_KERNARG_WARM
    constexpr int LINES = (BYTES + 63) / 64;
    const void *ka = (const void *)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned t[LINES];
    kernarg_touch_all<0, LINES>(ka, t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < LINES; ++i) asm volatile("" ::"s"(t[i]));      // (the destinations stay allocated until the loads are back)
#endif
}
// The same in two halves, for kernels whose FIRST loads go through pointers that are leading scalar kernel arguments: those
// arrive in SGPRs with the wave (kernarg preload, -mllvm -amdgpu-kernarg-preload-count=16 in build.py: up to 16 dwords of leading
// non-struct arguments), so the loads can leave while the rest of the segment is still on its way -- issue(), the first loads,
// wait().  (A struct argument ends the preloaded prefix: pointers inside Weights / Saved / ... never are.)
template <int BYTES>
struct KernargWarm {
    static constexpr int LINES = (BYTES + 63) / 64;
    unsigned t[LINES];
    __device__ __forceinline__ void issue() {
#ifndef TT_DBG_NO_KERNARG_WARM
        kernarg_touch_all<0, LINES>((const void *)__builtin_amdgcn_kernarg_segment_ptr(), t);
#endif
    }
    __device__ __forceinline__ void wait() {
#ifndef TT_DBG_NO_KERNARG_WARM
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < LINES; ++i) asm volatile("" ::"s"(t[i]));
#endif
    }
};
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// __restrict__ on the members: none of these buffers alias, and without it every store (saved activations,
// gradients) pins the loads that follow it in program order
struct Weights {
    const float *__restrict__ w1, *__restrict__ b1, *__restrict__ g1, *__restrict__ be1, *__restrict__ w2,
        *__restrict__ b2, *__restrict__ g2, *__restrict__ be2, *__restrict__ w3, *__restrict__ b3, *__restrict__ wa,
        *__restrict__ ba;
    const _Float16 *__restrict__ img;     // fc2 image (nullptr: products on the f32 MFMA from w2)
};
struct Saved {          // forward activations kept for the backward (all [B, .] row-major f32)
    float *__restrict__ xh1, *__restrict__ h1;    // [B,400] normalised fc1 output (before gamma/beta), post-ReLU activation
    float *__restrict__ xh2, *__restrict__ h2;    // [B,300] normalised fc2 output, post-ReLU (critic: after + action_value)
    float *__restrict__ rstd1, *__restrict__ rstd2;   // [B]
};

// Reductions on the DPP path (VALU speed) instead of __shfl_xor, which hipcc turns into ds_bpermute: an LDS round trip
// (~100 cycles) per step, and these sums sit on the critical path of every row phase.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row (= the lanes that share lane >> 4); every lane of the row gets it
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f<0xB1>(v);       // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_f<0x4E>(v);       // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_f<0x141>(v);      // row_half_mirror: the other quad of the half row
    v += dpp_f<0x140>(v);      // row_mirror: the other half row
    return v;
}

// sum over the wave; the result is wave-uniform
__device__ __forceinline__ float wave_sum64(float v) {
    const int u = __builtin_bit_cast(int, row_sum16(v));
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48)));
}

// max over the wave of a non-negative number; the result is wave-uniform
__device__ __forceinline__ float wave_max64(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    const int u = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48))));
}

// Workgroup barrier that publishes LDS only.  __syncthreads() also drains every outstanding GLOBAL store of the wave
// (s_waitcnt vmcnt(0) in front of s_barrier): the learn() kernels store their saved activations / per-row gradients right
// before their barriers, and each such barrier then cost a store round trip (~1-2 us) that nothing in the workgroup waits for.
// No launch here hands global data from one wave to another of the same workgroup; loads in flight stay tracked by the
// compiler (it waits at their first use).
__device__ __forceinline__ void lds_barrier() {
#ifdef TT_DBG_FULL_BARRIER
    __syncthreads();
    return;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// combine two per-wave, per-row partials (valid in every lane of the 16-lane group of that row) across the waves.
// red: [2][NW waves][16 rows].  Two barriers for both; every lane returns the totals of ITS four rows (r = 0..3).
__device__ __forceinline__ void cross_wave_sum2(float *red, int wave, int l4, int l15, float (&v)[4], float (&w2)[4]) {
    lds_barrier();                         // previous use of `red` is over
    if (l15 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[wave * TR + l4 * 4 + r] = v[r];
            red[NW * TR + wave * TR + l4 * 4 + r] = w2[r];
        }
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = l4 * 4 + r;
        float t = 0.f, u = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { t += red[w * TR + row]; u += red[NW * TR + w * TR + row]; }
        v[r] = t; w2[r] = u;
    }
}

// ------------------------------------------------------------------------------------------------------
// MFMA operand convention used by every product below.  v_mfma_f32_16x16x4_f32 sums over 4 values of k per
// instruction and lane l supplies k_local = l >> 4.  A dot product does not care in which ORDER k is visited, so one
// "k16 step" (16 consecutive k) is issued as 4 MFMAs where lane l contributes k = k0 + 4*(l>>4) + ks in MFMA ks:
// the four values a lane needs are then CONTIGUOUS in memory and arrive as one 16-byte load (global float4 or
// ds_read_b128) instead of four 4-byte ones.  Weights are read straight from L2 this way (no LDS ring, no
// barriers inside the K loops), so the loads of many steps can be in flight at once.
// The same trick on the N side ("column group"): lane l loads float4 W[k][c0 + 4*(l&15) .. +3] and uses component t
// as the B operand of output tile t, whose 16 columns are c0 + 4*(l&15) + t.

// Row phases (LayerNorms, head, their backward): a wave owns two rows and its lanes take the columns FOUR AT A TIME --
// columns 4 lane .. 4 lane + 3 and 256 + 4 lane .. + 3 -- so that every access of a row is a 16-byte one (LDS tile, saved
// activations, per-column vectors) and the two f16 planes leave as 8-byte stores: a quarter of the memory instructions of a
// lane-strided walk (column lane + 64 i), and no per-column branches.
constexpr int RV = 2;                                            // float4 groups of a row per lane
__device__ __forceinline__ int rv_col(const int lane, const int i) { return 4 * lane + 256 * i; }
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_ld(const float *p, const bool ok) {     // 16-byte aligned p; zeros where !ok
    return ok ? *reinterpret_cast<const float4 *>(p) : f4_zero();
}
__device__ __forceinline__ float4 f4_ldu(const float *p) { return *reinterpret_cast<const float4 *>(p); }     // 16-byte aligned p
__device__ __forceinline__ float f4_sum(const float4 v) { return (v.x + v.y) + (v.z + v.w); }
using h2v = __attribute__((ext_vector_type(2))) _Float16;
// four f32 -> their f16 pieces (round to nearest): h = rn16(x), m = rn16(x - h), as two 8-byte groups
__device__ __forceinline__ void split4(const float4 x, uint2 &ph, uint2 &pm) {
    h2v a, b, c, d;
    a[0] = (_Float16)x.x; a[1] = (_Float16)x.y; b[0] = (_Float16)x.z; b[1] = (_Float16)x.w;
    c[0] = (_Float16)(x.x - (float)a[0]); c[1] = (_Float16)(x.y - (float)a[1]);
    d[0] = (_Float16)(x.z - (float)b[0]); d[1] = (_Float16)(x.w - (float)b[1]);
    ph = make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
    pm = make_uint2(__builtin_bit_cast(uint32_t, c), __builtin_bit_cast(uint32_t, d));
}

// forward on a small batch.  out [B]: q (critic) or mu = tanh(.) (actor).  dq_da [B] (critic, optional):
// dQ/da = sum_j wq[j] * [z_j > 0] * wa[j], which is all of the critic the actor's gradient needs because the action
// enters after LayerNorm2 (networks.py:62-66).
// waves_per_eu(1,1): tell the scheduler NOT to trade the deep load pipelining for occupancy it cannot use anyway
// (16 workgroups on 256 CUs); without it hipcc keeps ~60 VGPRs and issues the weight loads a few at a time
struct EarlyW {
    const float *__restrict__ w1, *__restrict__ b1, *__restrict__ g1, *__restrict__ be1;
};
template <bool CRITIC, class Hook = NoHook>
__device__ __forceinline__ void fwd_small_body(const int n, const float *__restrict__ obs,
                                               const float *__restrict__ action, const Weights &W,
                                               float *__restrict__ out, const Saved &sv, float *__restrict__ dq_da,
                                               float *__restrict__ z_state, float *__restrict__ h1_s,
                                               float *__restrict__ z_s, float *__restrict__ w1_s, const int row0,
                                               const float *__restrict__ obs_row_lane = nullptr,
                                               const bool act_given = false, const float act_row0 = 0.f,
                                               const float act_row1 = 0.f, unsigned long long *dq_words = nullptr,
                                               const unsigned dq_epoch = 0u, const EarlyW *early = nullptr,
                                               const Hook &loads_issued = Hook()) {
    // early (optional): fc1 and the layer-1 vectors through pointers that reached the wave in SGPRs (leading kernel arguments) --
    // the same addresses as W's; loads_issued(): called once this phase's loads are out (the kernel's wait for the rest of its arguments)
    const EarlyW E = early ? *early : EarlyW{W.w1, W.b1, W.g1, W.be1};
    // obs_row_lane (optional): this lane's observation row for layer 1 (row row0 + (lane & 15)) when the rows are gathered
    // from a replay ring instead of read from obs [n,23]; act_given / act_row0, 1: the actions of this wave's two rows likewise.
    // h1_s [16][404]: fc1 pre-activations, then the A operand of layer 2; z_s [16][308]: fc2 pre-activations.
    // The two products split the COLUMNS over the 8 waves; everything per row (both LayerNorms, the head) is done by the
    // wave that owns the row (wave w: rows 2w, 2w+1; lanes stride the columns) after ONE hand-over through LDS, with
    // wave-level reductions: 4 barriers per forward instead of 13, and the saved activations leave as whole rows.
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;

    STAMP(0);
    constexpr int MT1 = (NT1 + NW - 1) / NW, MT2 = (NT2 + NW - 1) / NW;
    // every per-column vector this lane will need, loaded NOW: the uses sit behind barriers, which the compiler cannot
    // move a load across, so each would cost an exposed L2 round trip there
    // (layer 1's vectors here; layer 2's are requested where layer 2 starts and land behind its products)
    float4 pb1[RV], pg1[RV], pbe1[RV], pb2[RV], pg2[RV], pbe2[RV], pw3[RV], pwa[RV], pba[RV];
    // EVERY load of this phase is unconditional, from a clamped address, and issued before the first MFMA (a value that must be
    // zero is zeroed afterwards).  Guarded loads (`k < IN ? w1[..] : 0`) made the compiler wrap each in a saved exec mask and emit
    // layer 1 tile by tile -- loads, wait, six MFMAs, the next tile's loads ...: four dependent round trips to L2 for a K = 23
    // product (3.8 us of the row kernel; round 3's stamps), and one more for the second group of per-column vectors.
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        const int c = rv_col(lane, i), cc = c < H1 ? c : 0;       // (values beyond column 399 are never used)
        pb1[i] = *reinterpret_cast<const float4 *>(E.b1 + cc); pg1[i] = *reinterpret_cast<const float4 *>(E.g1 + cc);
        pbe1[i] = *reinterpret_cast<const float4 *>(E.be1 + cc);
    }
    // ---- layer 1 (K = 23): operands straight from global; this wave's column tiles t = wave, wave+NW, ...
    f32x4 acc1[MT1];
#pragma unroll
    for (int i = 0; i < MT1; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int GU_F = 3;
    const bool img = W.img != nullptr;                     // (uniform over the launch)
    const int nt_f = (NT2 - wave + NW - 1) / NW;          // layer-2 tiles of this wave: 3 (waves 0..3) or 2
    f16x8 bpre[GU_F][MT2][2];
    {
        // fc1 [400,23] reaches the workgroup ONCE, as 2300 coalesced float4 (36.8 KB), through LDS: read in place -- each lane its
        // six values of each of its wave's tiles, rows 92 bytes apart -- it cost 24 four-byte loads per lane that touch 12-16
        // cache lines each, and the CU's one address pipe was busy with them for ~2 us (8 waves x 36 such loads): layer 1 of a
        // K = 23 product took 3.1-3.8 us.  The observation rows (6 loads per lane) are still read in place.
        float a[INP / 4];
        const float *arow = obs_row_lane ? obs_row_lane : obs + (size_t)min(row0 + l15, n - 1) * IN;
        const bool arow_ok = row0 + l15 < n;
        constexpr int W1_F4 = H1 * IN / 4, W1_PER = (W1_F4 + 64 * NW - 1) / (64 * NW);
        float4 wv[W1_PER];
#pragma unroll
        for (int q = 0; q < W1_PER; ++q) wv[q] = f4_ldu(E.w1 + 4 * min(tid + 64 * NW * q, W1_F4 - 1));
#pragma unroll
        for (int ks = 0; ks < INP / 4; ++ks) a[ks] = arow[min(ks * 4 + l4, IN - 1)];
        loads_issued();
        SUB(0);
#pragma unroll
        for (int q = 0; q < W1_PER; ++q)
            if (tid + 64 * NW * q < W1_F4) *reinterpret_cast<float4 *>(w1_s + 4 * (tid + 64 * NW * q)) = wv[q];
        SUB(1);
        lds_barrier();
        SUB(2);
#pragma unroll
        for (int ks = 0; ks < INP / 4; ++ks) a[ks] = (ks * 4 + l4 < IN && arow_ok) ? a[ks] : 0.f;
        // every fragment of the wave's tiles first (24 LDS reads in flight together), then the products k-step by k-step ACROSS the
        // tiles: six MFMAs into one accumulator in a row wait for each other (a dependent 16x16x4 issues every ~40 cycles, an
        // independent one every 32), tile by tile that was the order
        float b[MT1][INP / 4];
#pragma unroll
        for (int i = 0; i < MT1; ++i) {
            const float *wr = w1_s + (min(wave + NW * i, NT1 - 1) * 16 + l15) * IN;
#pragma unroll
            for (int ks = 0; ks < INP / 4; ++ks) {
                const float v = wr[min(ks * 4 + l4, IN - 1)];
                b[i][ks] = ks * 4 + l4 < IN ? v : 0.f;          // (only k = 23, the padding of the last k4 group, is not)
            }
        }
#pragma unroll
        for (int ks = 0; ks < INP / 4; ++ks)
#pragma unroll
            for (int i = 0; i < MT1; ++i)
                if (wave + NW * i < NT1) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[i][ks], acc1[i], 0, 0, 0);
        // The first group of fc2 fragments (three k32 steps of this wave's tiles, straight from L2) is requested HERE: it does not
        // depend on the activations, and requested where layer 2 starts it cost that phase one exposed round trip.  (Requested
        // in front of layer 1 it delays the other waves' layer-1 operands: 144 KB through the same address pipe.)
        if (img) {
#pragma unroll
            for (int u = 0; u < GU_F; ++u)
#pragma unroll
                for (int i = 0; i < MT2; ++i) {
                    const _Float16 *bq = W.img + ((size_t)min(wave + NW * i, NT2 - 1) * FW_STEPS * 64 + lane) * 8 + 512 * u;
                    bpre[u][i][0] = *reinterpret_cast<const f16x8 *>(bq);
                    bpre[u][i][1] = *reinterpret_cast<const f16x8 *>(bq + IMG_FWD);
                }
        }
    }
    STAMP(1);
    // accumulator element [i][r] is row l4*4 + r, column (wave + NW*i)*16 + l15
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        const int t = wave + NW * i;
        if (t < NT1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h1_s[(l4 * 4 + r) * HS1 + t * 16 + l15] = acc1[i][r];
        }
    }
    SUB(3);
    lds_barrier();
    SUB(4);
    // bias, LayerNorm(400) (biased variance, eps 1e-5), ReLU for this wave's two rows.  With an fc2 image the rows leave as
    // the two f16 planes of the layer-2 operand, which share the buffer with the f32 tile: every wave has read its rows
    // before any plane is written
    _Float16 *const ap_s = reinterpret_cast<_Float16 *>(h1_s);          // [2 planes][16][HSH]
    float4 xr[TR / NW][RV];
#pragma unroll
    for (int rr = 0; rr < TR / NW; ++rr) {
        const int lr = wave * (TR / NW) + rr;
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i);
            const bool ok = c < H1;
            const float4 t = *reinterpret_cast<const float4 *>(&h1_s[lr * HS1 + (ok ? c : 0)]);
            xr[rr][i] = ok ? make_float4(t.x + pb1[i].x, t.y + pb1[i].y, t.z + pb1[i].z, t.w + pb1[i].w) : f4_zero();
        }
    }
    if (img) lds_barrier();
    SUB(5);
#pragma unroll
    for (int rr = 0; rr < TR / NW; ++rr) {
        const int lr = wave * (TR / NW) + rr, row = row0 + lr;
        const float mean = wave_sum64(f4_sum(xr[rr][0]) + f4_sum(xr[rr][1])) * (1.f / H1);
        float4 dv[RV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const bool ok = rv_col(lane, i) < H1;
            const float4 x = xr[rr][i];
            dv[i] = ok ? make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean) : f4_zero();
            ss = fmaf(dv[i].x, dv[i].x, ss); ss = fmaf(dv[i].y, dv[i].y, ss);
            ss = fmaf(dv[i].z, dv[i].z, ss); ss = fmaf(dv[i].w, dv[i].w, ss);
        }
        const float rstd = rsqrtf(wave_sum64(ss) * (1.f / H1) + 1e-5f);
        if (sv.rstd1 && lane == 0 && row < n) sv.rstd1[row] = rstd;
        const bool save = sv.xh1 && row < n;
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i);
            if (c < H1) {
                const float4 xh = make_float4(dv[i].x * rstd, dv[i].y * rstd, dv[i].z * rstd, dv[i].w * rstd);
                const float4 h = make_float4(fmaxf(fmaf(xh.x, pg1[i].x, pbe1[i].x), 0.f), fmaxf(fmaf(xh.y, pg1[i].y, pbe1[i].y), 0.f),
                                             fmaxf(fmaf(xh.z, pg1[i].z, pbe1[i].z), 0.f), fmaxf(fmaf(xh.w, pg1[i].w, pbe1[i].w), 0.f));
                if (img) {
                    uint2 ph, pm;
                    split4(make_float4(h.x * SXL, h.y * SXL, h.z * SXL, h.w * SXL), ph, pm);
                    *reinterpret_cast<uint2 *>(ap_s + lr * HSH + c) = ph;
                    *reinterpret_cast<uint2 *>(ap_s + TR * HSH + lr * HSH + c) = pm;
                } else {
                    *reinterpret_cast<float4 *>(&h1_s[lr * HS1 + c]) = h;
                }
                if (save) {
                    *reinterpret_cast<float4 *>(sv.xh1 + (size_t)row * H1 + c) = xh;
                    *reinterpret_cast<float4 *>(sv.h1 + (size_t)row * H1 + c) = h;
                }
            } else if (img && c < K2P) {                   // K padding of the planes (inputs 400..415)
                *reinterpret_cast<uint2 *>(ap_s + lr * HSH + c) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2 *>(ap_s + TR * HSH + lr * HSH + c) = make_uint2(0u, 0u);
            }
        }
    }
    SUB(6);
    lds_barrier();   // the 16 x 400 activation tile is complete
    STAMP(2);
    // layer 2's per-column vectors: requested behind the LAST group of fc2 fragments (the prefetch registers are free by then),
    // so that they land behind that group's products
    auto load_l2_vectors = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            // (unconditional, from a clamped column: a guarded load costs a saved exec mask and, worse, its own wait -- see layer 1;
            // every use of these values is guarded by c < H2)
            const int c = rv_col(lane, i), cc = c < H2 ? c : 0;
            pb2[i] = f4_ldu(W.b2 + cc); pg2[i] = f4_ldu(W.g2 + cc); pbe2[i] = f4_ldu(W.be2 + cc);
            pw3[i] = f4_ldu(W.w3 + cc);
            if (CRITIC) { pwa[i] = f4_ldu(W.wa + cc); pba[i] = f4_ldu(W.ba + cc); }
            else { pwa[i] = f4_zero(); pba[i] = f4_zero(); }
        }
    };

    // ---- layer 2: this wave's column tiles t = wave, wave+NW, ... (3 or 2 of the 20); A from the LDS tile (one
    // ds_read_b128 per k16 step), B = fc2 rows straight from L2 (one float4 per tile per k16 step), k visited in
    // the permuted order described above
    f32x4 acc2[MT2];
#pragma unroll
    for (int i = 0; i < MT2; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wrow[MT2];
    bool wreal[MT2];
#pragma unroll
    for (int i = 0; i < MT2; ++i) {
        const int nn = (wave + NW * i) * 16 + l15;
        wreal[i] = wave + NW * i < NT2 && nn < H2;
        wrow[i] = W.w2 + (size_t)(wreal[i] ? nn : 0) * H1 + 4 * l4;
    }
    if (img) {
        // A = the planes (one ds_read_b128 per plane and k32 step), B = this wave's rows of the image's [n][k] half straight
        // from L2 (16 B per plane, tile and step), three MFMAs per tile and step, small terms first; the fragments of the
        // NEXT three steps are requested before the current three issue
        const int nt = nt_f;
        const _Float16 *ah = ap_s + l15 * HSH + 8 * l4;
        const _Float16 *bh[MT2];
#pragma unroll
        for (int i = 0; i < MT2; ++i) bh[i] = W.img + ((size_t)min(wave + NW * i, NT2 - 1) * FW_STEPS * 64 + lane) * 8;
        constexpr int NS = FW_STEPS, GU = 3, NGRP = (NS + GU - 1) / GU;
        f16x8 bcur[GU][MT2][2], bnxt[GU][MT2][2];
        auto load_group = [&](const int g, f16x8 (&dst)[GU][MT2][2]) {
#pragma unroll
            for (int u = 0; u < GU; ++u)
#pragma unroll
                for (int i = 0; i < MT2; ++i) {
                    const int c = g * GU + u;
                    if (c < NS && i < nt) {
                        dst[u][i][0] = *reinterpret_cast<const f16x8 *>(bh[i] + 512 * c);
                        dst[u][i][1] = *reinterpret_cast<const f16x8 *>(bh[i] + IMG_FWD + 512 * c);
                    }
                }
        };
        static_assert(GU == GU_F, "the prefetched group is group 0");
        // Two fragment sets that swap roles from group to group (the loop is unrolled: no copies).  Handing the next set over with
        // `bcur = bnxt` cost ~460 register moves per wave in this phase -- as many issue cycles as its MFMAs.
        auto products = [&](const int g, const f16x8 (&b)[GU][MT2][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int c = g * GU + u;
                if (c < NS) {
                    const f16x8 a_h = *reinterpret_cast<const f16x8 *>(ah + 32 * c);
                    const f16x8 a_m = *reinterpret_cast<const f16x8 *>(ah + TR * HSH + 32 * c);
#pragma unroll
                    for (int i = 0; i < MT2; ++i) if (i < nt) acc2[i] = mfma_h(a_m, b[u][i][0], acc2[i]);
#pragma unroll
                    for (int i = 0; i < MT2; ++i) if (i < nt) acc2[i] = mfma_h(a_h, b[u][i][1], acc2[i]);
#pragma unroll
                    for (int i = 0; i < MT2; ++i) if (i < nt) acc2[i] = mfma_h(a_h, b[u][i][0], acc2[i]);
                }
            }
        };
        auto step = [&](const int g, const f16x8 (&use)[GU][MT2][2], f16x8 (&fill)[GU][MT2][2]) __attribute__((always_inline)) {
            if (g + 1 < NGRP) load_group(g + 1, fill);
            else load_l2_vectors();
            __builtin_amdgcn_sched_barrier(0);
            products(g, use);
            __builtin_amdgcn_sched_barrier(0);
        };
        (void)bcur;
#pragma unroll
        for (int g = 0; g < NGRP; g += 2) {
            if (g == 0) step(0, bpre, bnxt);
            else step(g, bcur, bnxt);
            if (g + 1 < NGRP) step(g + 1, bnxt, bcur);
        }
#pragma unroll
        for (int i = 0; i < MT2; ++i) acc2[i] *= UNSC_L;
    } else {
    load_l2_vectors();
    const float *arow = h1_s + l15 * HS1 + 4 * l4;
#pragma unroll 5
    for (int c = 0; c < H1 / 16; ++c) {
        const float4 av = *reinterpret_cast<const float4 *>(arow + 16 * c);
        float4 bv[MT2];
#pragma unroll
        for (int i = 0; i < MT2; ++i) {
            bv[i] = *reinterpret_cast<const float4 *>(wrow[i] + 16 * c);
            if (!wreal[i]) bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ks-major: consecutive MFMAs write different accumulators (dependent latency 40 > issue 32 cycles)
#pragma unroll
        for (int i = 0; i < MT2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[i].x, acc2[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[i].y, acc2[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[i].z, acc2[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[i].w, acc2[i], 0, 0, 0);
    }
    }

    STAMP(3);
    // ---- hand the fc2 pre-activations over (columns 300..319 of the last tile are padding)
#pragma unroll
    for (int i = 0; i < MT2; ++i) {
        const int col = (wave + NW * i) * 16 + l15;
        if (wave + NW * i < NT2 && col < H2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) z_s[(l4 * 4 + r) * DS + col] = acc2[i][r];
        }
    }
    lds_barrier();
    // ---- epilogue for this wave's two rows: bias, LayerNorm(300), (critic: + action_value(a)), ReLU, head
    float avs[TR / NW];                                    // loaded before the first row's stores (see k_bwd_rows, phase A)
#pragma unroll
    for (int rr = 0; rr < TR / NW; ++rr) {
        const int row = row0 + wave * (TR / NW) + rr;
        avs[rr] = (CRITIC && !z_state && row < n) ? (act_given ? (rr == 0 ? act_row0 : act_row1) : action[row]) : 0.f;
    }
    const float b3 = W.b3[0];
#pragma unroll
    for (int rr = 0; rr < TR / NW; ++rr) {
        const int lr = wave * (TR / NW) + rr, row = row0 + lr;
        float4 x[RV];
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i);
            const bool ok = c < H2;
            const float4 t = *reinterpret_cast<const float4 *>(&z_s[lr * DS + (ok ? c : 0)]);
            x[i] = ok ? make_float4(t.x + pb2[i].x, t.y + pb2[i].y, t.z + pb2[i].z, t.w + pb2[i].w) : f4_zero();
        }
        const float mean = wave_sum64(f4_sum(x[0]) + f4_sum(x[1])) * (1.f / H2);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const bool ok = rv_col(lane, i) < H2;
            x[i] = ok ? make_float4(x[i].x - mean, x[i].y - mean, x[i].z - mean, x[i].w - mean) : f4_zero();      // deviations
            ss = fmaf(x[i].x, x[i].x, ss); ss = fmaf(x[i].y, x[i].y, ss); ss = fmaf(x[i].z, x[i].z, ss); ss = fmaf(x[i].w, x[i].w, ss);
        }
        const float rstd = rsqrtf(wave_sum64(ss) * (1.f / H2) + 1e-5f);
        if (sv.rstd2 && lane == 0 && row < n) sv.rstd2[row] = rstd;
        if (CRITIC && z_state) {
            // state branch only (networks.py:55-61): z_state [B,300] = bn2(fc2(relu(bn1(fc1(s))))) before the action
            // enters; the TD prologue of the critic's backward (or k_head_td) finishes q once the action is known, so
            // this pass can run NEXT TO the actor pass that produces it
#pragma unroll
            for (int i = 0; i < RV; ++i) {
                const int c = rv_col(lane, i);
                if (c < H2 && row < n)
                    *reinterpret_cast<float4 *>(z_state + (size_t)row * H2 + c) =
                        make_float4(fmaf(x[i].x * rstd, pg2[i].x, pbe2[i].x), fmaf(x[i].y * rstd, pg2[i].y, pbe2[i].y),
                                    fmaf(x[i].z * rstd, pg2[i].z, pbe2[i].z), fmaf(x[i].w * rstd, pg2[i].w, pbe2[i].w));
            }
            continue;
        }
        const float av = avs[rr];
        const bool save = sv.xh2 && row < n;
        float dot = 0.f, dqa = 0.f;
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i);
            if (c < H2) {
                const float xhv[4] = {x[i].x * rstd, x[i].y * rstd, x[i].z * rstd, x[i].w * rstd};
                const float gv[4] = {pg2[i].x, pg2[i].y, pg2[i].z, pg2[i].w}, bev[4] = {pbe2[i].x, pbe2[i].y, pbe2[i].z, pbe2[i].w};
                const float w3v[4] = {pw3[i].x, pw3[i].y, pw3[i].z, pw3[i].w};
                const float wav[4] = {pwa[i].x, pwa[i].y, pwa[i].z, pwa[i].w}, bav[4] = {pba[i].x, pba[i].y, pba[i].z, pba[i].w};
                float hv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float z = fmaf(xhv[q], gv[q], bev[q]);
                    if (CRITIC) z += fmaf(av, wav[q], bav[q]);
                    hv[q] = fmaxf(z, 0.f);
                    dot = fmaf(hv[q], w3v[q], dot);
                    if (CRITIC) dqa = fmaf(z > 0.f ? w3v[q] : 0.f, wav[q], dqa);
                }
                if (save) {
                    *reinterpret_cast<float4 *>(sv.xh2 + (size_t)row * H2 + c) = make_float4(xhv[0], xhv[1], xhv[2], xhv[3]);
                    *reinterpret_cast<float4 *>(sv.h2 + (size_t)row * H2 + c) = make_float4(hv[0], hv[1], hv[2], hv[3]);
                }
            }
        }
        dot = wave_sum64(dot);
        if (CRITIC && dq_da) dqa = wave_sum64(dqa);
        if (lane == 0 && row < n) {
            const float v = dot + b3;
            out[row] = CRITIC ? v : tanhf(v);
            if (CRITIC && dq_da) {
                dq_da[row] = dqa;
                // k_actor_tail: read by workgroups of the SAME launch on other XCDs.  Value and learn step leave as ONE 8-byte atomic
                // store (agent scope: coherent by itself), so a reader that finds the step it waits for has the value of that step
                // -- no second word whose store could overtake the value's on its way to memory, hence no release fence.
                if (dq_words)
                    __hip_atomic_store(dq_words + row, ((unsigned long long)dq_epoch << 32) | (unsigned long long)__float_as_uint(dqa),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    STAMP(4);
}

template <bool CRITIC>
__global__ __launch_bounds__(64 * NW) void k_fwd_small(const int n, const float *__restrict__ obs,
                                                   const float *__restrict__ action, const float *__restrict__ w1e,
                                                   const float *__restrict__ b1e, const float *__restrict__ g1e,
                                                   const float *__restrict__ be1e, const Weights W,
                                                   float *__restrict__ out, const Saved sv, float *__restrict__ dq_da,
                                                   float *__restrict__ z_state) {
    // (n .. be1e: 14 dwords of leading scalar arguments = the preloaded prefix; w1e .. be1e repeat W.w1, b1, g1, be1)
    __shared__ __attribute__((aligned(16))) float h1_s[H1S_FLOATS];
    __shared__ __attribute__((aligned(16))) float z_s[TR * DS];
    __shared__ __attribute__((aligned(16))) float w1_s[H1 * IN];
    KernargWarm<64 + (int)sizeof(Weights) + 8 + (int)sizeof(Saved) + 16> warm;
    warm.issue();
    KBEGIN(3);
    const EarlyW E{w1e, b1e, g1e, be1e};
    auto hook = [&]() __attribute__((always_inline)) { warm.wait(); };
    fwd_small_body<CRITIC, decltype(hook)>(n, obs, action, W, out, sv, dq_da, z_state, h1_s, z_s, w1_s, blockIdx.x * TR, nullptr, false, 0.f,
                                           0.f, nullptr, 0u, &E, hook);
    KEND(3);
}

// Up to four independent forwards on the same number of rows in ONE launch (workgroup b serves job b / blocks_per_job):
// learn()'s first phase -- target actor on s', the target critic's state branch on s', Q(s,a) and mu(s) -- needs no
// stream fork/join inside the captured graph this way (each fork costs 10-20 us of cross-queue signalling, as much as
// the kernel it would hide).
struct FwdJob {
    const float *obs, *action;
    Weights W;
    float *out;
    Saved sv;
    float *dq_da, *z_state;
    int critic;
};
struct FwdJobs {
    FwdJob j[4];
    int n, blocks_per_job;
    // sampled = 1: the launch makes the replay draw R itself (tt_mlp_forward_multi_sampled) -- every workgroup finds the ring
    // rows of ITS 16 batch rows (the same Philox draw in every job) and reads s / s' / a straight from the ring; the
    // workgroups of job write_s also leave s and a, those of job write_s2 leave s', r and done in the draw's batch buffers for
    // the launches that follow (one launch and one dependent boundary less per learn() than tt_ring_sample + this)
    int sampled, write_s, write_s2;
    ttnet::RingSample R;
    long long *k_snapshot;       // (sampled) *R.k_dev as this launch saw it, for a later launch (tt_image_job) or nullptr
};
__global__ __launch_bounds__(64 * NW) void k_fwd_multi(const FwdJobs J) {
    __shared__ __attribute__((aligned(16))) float h1_s[H1S_FLOATS];
    __shared__ __attribute__((aligned(16))) float z_s[TR * DS];
    __shared__ __attribute__((aligned(16))) float w1_s[H1 * IN];
    kernarg_warm<(int)sizeof(FwdJobs)>();
    const int job = blockIdx.x / J.blocks_per_job, row0 = (blockIdx.x - job * J.blocks_per_job) * TR;
    const FwdJob &q = J.j[job];
    KBEGIN(0);
#ifdef TT_STAMPS   // the last end of the PREVIOUS learn()'s last launch, before this learn() overwrites anything: [5][0][0]
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        unsigned long long m = 0;
        for (int i = threadIdx.x; i < 512; i += 64) m = max(m, g_kblk[4][i][1]);
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned long long)__shfl_xor((long long)m, o));
        if (threadIdx.x == 0) g_kblk[5][0][0] = m;
    }
#endif
    static_assert(TR / NW == 2, "two rows per wave");
    const float *orow = nullptr;
    bool have_act = false;
    float act_r0 = 0.f, act_r1 = 0.f;
    if (J.sampled) {
        ttnet::await_progress(J.R.progress, J.R.k_dev);
        const int tid = threadIdx.x, wave = tid >> 6, l15 = tid & 15;
        if (J.k_snapshot && blockIdx.x == 0 && tid == 0) *J.k_snapshot = *J.R.k_dev;
        const bool from_s = q.obs == J.R.s_out;                  // this job reads s (else s')
        {
            const ttnet::RingPick p = ttnet::ring_sample_index(J.R, min(row0 + l15, J.n - 1));
            orow = from_s ? ttnet::ring_pick_s(J.R, p) : ttnet::ring_pick_s2(J.R, p);
        }
        if (q.critic && q.action) {
            act_r0 = ttnet::ring_pick_a(J.R, ttnet::ring_sample_index(J.R, min(row0 + wave * 2, J.n - 1)));
            act_r1 = ttnet::ring_pick_a(J.R, ttnet::ring_sample_index(J.R, min(row0 + wave * 2 + 1, J.n - 1)));
            have_act = true;
        }
        if (job == J.write_s || job == J.write_s2) {
            // the batch rows of this workgroup for the later launches: thread i < 16 x 23 copies one feature
            // (loads here and the stores after the forward -- so that its own loads do not queue behind this copy -- changed
            // nothing: 18.2 us for the slowest workgroup either way)
            const int lr = tid / ttnet::IN, c = tid - lr * ttnet::IN, b = row0 + lr;
            if (lr < TR && b < J.n) {
                const ttnet::RingPick p = ttnet::ring_sample_index(J.R, b);
                if (job == J.write_s) {
                    J.R.s_out[(size_t)b * ttnet::IN + c] = ttnet::ring_pick_s(J.R, p)[c];
                    if (c == 0) {
                        J.R.a_out[b] = ttnet::ring_pick_a(J.R, p);
                        if (J.R.idx_out) { J.R.idx_out[2 * b] = p.side ? -1 : p.t; J.R.idx_out[2 * b + 1] = p.side ? p.j : p.e; }
                    }
                }
                if (job == J.write_s2) {
                    J.R.s2_out[(size_t)b * ttnet::IN + c] = ttnet::ring_pick_s2(J.R, p)[c];
                    if (c == 0) { J.R.r_out[b] = ttnet::ring_pick_r(J.R, p); J.R.d_out[b] = ttnet::ring_pick_d(J.R, p); }
                }
            }
        }
    }
    if (q.critic)
        fwd_small_body<true>(J.n, q.obs, q.action, q.W, q.out, q.sv, q.dq_da, q.z_state, h1_s, z_s, w1_s, row0, orow, have_act, act_r0, act_r1);
    else
        fwd_small_body<false>(J.n, q.obs, q.action, q.W, q.out, q.sv, nullptr, nullptr, h1_s, z_s, w1_s, row0, orow);
    KEND(0);
}

// ------------------------------------------------------------------------------------------------------
// per-row backward of one net.  16 rows per workgroup.
//   mode 0: d_out[b] is given (gradient w.r.t. `out`)
//   mode 1: d_out[b] = scale * (out[b] - y[b])          critic MSE: d/dq mean((y - q)^2), scale = 2/B
//   mode 2: d_out[b] = scale * aux[b]                    actor: d/dmu mean(-Q), aux = dQ/da, scale = -1/B
//   mode 3: unit gradient 1 at the head's PRE-activation of every row (no tanh factor): the per-row gradients of a row are
//           linear in that number, so the real ones are these times the row's d(loss)/d(pre) -- applied by k_bwd_weights
// actor (CRITIC = false): `out` is mu = tanh(pre) and the head gradient is d_out * (1 - mu^2).
// Writes dpre [B], dz [B,300] (grad at the ReLU-masked LayerNorm2 output), dx2 [B,300] (grad at fc2's output),
// dy1 [B,400] (grad at the ReLU-masked LayerNorm1 output), dx1 [B,400] (grad at fc1's output).
constexpr int NG = 7;                         // 64-column groups covering the 400 columns of dH1 (the last is partial)

struct BwdOut {
    float *__restrict__ dpre, *__restrict__ dz, *__restrict__ dx2, *__restrict__ dy1, *__restrict__ dx1;
};


// Optional prologue of the critic's backward: the rest of the TARGET critic once the target actor's action is known
// (networks.py:62-68) and the TD target (DDPG_agent.py:89-93), for the rows this workgroup owns -- what k_head_td does
// as a launch of its own.  q'[b] = q(relu(z_state[b] + action_value(mu'[b]))), y[b] = r[b] + gamma q'[b] (1 - done[b]).
struct TdIn {
    const float *__restrict__ z_state, *__restrict__ mu_t, *__restrict__ r;
    const uint8_t *__restrict__ done;
    const float *__restrict__ wa, *__restrict__ ba, *__restrict__ w3, *__restrict__ b3;     // the target critic's
    float gamma;
    float *__restrict__ y_out, *__restrict__ q_out;
    long long *__restrict__ step_dev, *__restrict__ window_dev;
    float *__restrict__ bc_out;        // Adam's bias corrections of the new step (tt_td_input.bias_corr_out) or nullptr
    float beta1, beta2;
    int separate_tick;                 // 1: the counters are advanced by clock_tick() on a workgroup of its own, not by block 0
};

// Advance the learn-step (and sampling-window) counter; leave torch.optim.Adam's bias corrections of the new step: two f64
// pow() here, once, instead of in every thread of the two optimizer launches (~1.1 us on each launch's critical path).
__device__ inline void clock_tick(const TdIn &td) {
    long long t = 0;
    if (td.step_dev) { t = *td.step_dev + 1; *td.step_dev = t; }
    if (td.window_dev) *td.window_dev += 1;
    if (td.bc_out && td.step_dev) {
        td.bc_out[1] = td.beta1; td.bc_out[2] = td.beta2;
        td.bc_out[3] = (float)(1.0 - pow((double)td.beta1, (double)t));
        td.bc_out[4] = (float)(1.0 - pow((double)td.beta2, (double)t));
        td.bc_out[0] = __int_as_float((int)t);
    }
}

template <bool CRITIC>
__device__ __forceinline__ void bwd_rows_body(const int n, const int mode, const float scale,
                                              const float *__restrict__ d_out, const float *__restrict__ out,
                                              const float *__restrict__ y, const float *__restrict__ aux,
                                              const Weights &W, const Saved &sv, const BwdOut &o, const TdIn &td,
                                              float *__restrict__ dx2_s, float *__restrict__ red, float *__restrict__ rsc_s,
                                              const int row0) {
    // dx2_s [16][308]: A operand of phase B (with an fc2 image: its two f16 planes [2][16][328], each row scaled by a power
    // of two whose inverse / 64 goes to rsc_s [16]); red [2][NW][16]: cross-wave reductions
    const bool img = W.img != nullptr;                     // (uniform over the launch)
    _Float16 *const dxp_s = reinterpret_cast<_Float16 *>(dx2_s);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    STAMP(8);
    // ---- phase A: head, ReLU and LayerNorm2 backward; wave w owns rows 2w, 2w+1; a lane takes the 300 columns four at a
    // time (see RV).  Every load of both rows is issued before the first store: the outputs are plain pointers inside a struct,
    // so the compiler must assume a store may alias a later load and would otherwise serialise the two rows' round trips.
    constexpr int RPW = TR / NW;
    float4 w3c[RV], g2c[RV], h2v[RPW][RV], xh[RPW][RV], zt[RPW][RV], wat[RV], bat[RV], w3t[RV];
    float rs[RPW], gin[RPW], yin[RPW], outv[RPW], mut[RPW], rt[RPW], b3t = 0.f;
    bool dt[RPW];
    const bool with_td = CRITIC && td.z_state;               // (uniform over the launch)
    // Every load below is unconditional, from a clamped row / column (a value that must be zero is zeroed afterwards; most uses
    // are guarded anyway): guarded loads made the compiler emit this phase as a chain of exec-masked blocks, each with its own
    // wait -- five dependent round trips to L2 in front of the first arithmetic (5.0 us for this phase in round 3's stamps).
    // `with_td` is uniform over the launch: a scalar branch around loads that would dereference null pointers.
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        const int c = rv_col(lane, i), cc = c < H2 ? c : 0;
        w3c[i] = f4_ldu(W.w3 + cc); g2c[i] = f4_ldu(W.g2 + cc);
        wat[i] = f4_zero(); bat[i] = f4_zero(); w3t[i] = f4_zero();
    }
    if (with_td) {
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i), cc = c < H2 ? c : 0;
            wat[i] = f4_ldu(td.wa + cc); bat[i] = f4_ldu(td.ba + cc); w3t[i] = f4_ldu(td.w3 + cc);
        }
        b3t = td.b3[0];
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = row0 + wave * RPW + rr, rowc = min(row, n - 1);
        rs[rr] = sv.rstd2[rowc];
        outv[rr] = (mode != 0 || !CRITIC) ? out[rowc] : 0.f;
        gin[rr] = mode == 0 ? d_out[rowc] : (mode == 2 ? aux[rowc] : 0.f);
        yin[rr] = (mode == 1 && !with_td) ? y[rowc] : 0.f;
        mut[rr] = 0.f; rt[rr] = 0.f; dt[rr] = false;
        if (with_td) { mut[rr] = td.mu_t[rowc]; rt[rr] = td.r[rowc]; dt[rr] = td.done[rowc] != 0; }
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i);
            const size_t q = (size_t)rowc * H2 + (c < H2 ? c : 0);
            h2v[rr][i] = f4_ldu(sv.h2 + q);
            xh[rr][i] = f4_ldu(sv.xh2 + q);
            zt[rr][i] = with_td ? f4_ldu(td.z_state + q) : f4_zero();
        }
    }
#pragma unroll
    for (int i = 0; i < RV; ++i)                               // the TD dot product runs over every lane's columns: none beyond 299
        if (!(rv_col(lane, i) < H2)) w3t[i] = f4_zero();
    if (with_td && blockIdx.x == 0 && tid == 0 && !td.separate_tick) {
        if (td.step_dev) *td.step_dev += 1;
        if (td.window_dev) *td.window_dev += 1;      // a pipelined loop's sampling window moves on (read by LATER launches only)
    }
    // phase B's first group of fc2 fragments (two k32 steps of this wave's four tiles, straight from L2) is requested HERE,
    // behind phase A's own loads: it depends on nothing this kernel computes
    constexpr int GU_B = 2;
    f16x8 bpre[GU_B][4][2];
    if (img && wave < NG) {
        const _Float16 *bh0 = W.img + 2 * IMG_FWD + ((size_t)wave * 4 * BW_STEPS * 64 + lane) * 8;
#pragma unroll
        for (int u = 0; u < GU_B; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bpre[u][t][0] = *reinterpret_cast<const f16x8 *>(bh0 + (size_t)(t * BW_STEPS + u) * 512);
                bpre[u][t][1] = *reinterpret_cast<const f16x8 *>(bh0 + IMG_T + (size_t)(t * BW_STEPS + u) * 512);
            }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int lr = wave * RPW + rr, row = row0 + lr;
        const bool ok = row < n;
        float y_td = 0.f;
        if (with_td) {
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < RV; ++i) {
                dot = fmaf(fmaxf(zt[rr][i].x + fmaf(mut[rr], wat[i].x, bat[i].x), 0.f), w3t[i].x, dot);
                dot = fmaf(fmaxf(zt[rr][i].y + fmaf(mut[rr], wat[i].y, bat[i].y), 0.f), w3t[i].y, dot);
                dot = fmaf(fmaxf(zt[rr][i].z + fmaf(mut[rr], wat[i].z, bat[i].z), 0.f), w3t[i].z, dot);
                dot = fmaf(fmaxf(zt[rr][i].w + fmaf(mut[rr], wat[i].w, bat[i].w), 0.f), w3t[i].w, dot);
            }
            const float q = wave_sum64(dot) + b3t;
            y_td = dt[rr] ? rt[rr] : fmaf(td.gamma, q, rt[rr]);
            if (ok && lane == 0) {
                td.y_out[row] = y_td;
                if (td.q_out) td.q_out[row] = q;
            }
        }
        float dpre = 0.f;
        if (ok) {
            float g = mode == 0 ? gin[rr] : (mode == 1 ? scale * (outv[rr] - (with_td ? y_td : yin[rr])) : scale * gin[rr]);
            if (!CRITIC) g *= (1.f - outv[rr] * outv[rr]);
            if (mode == 3) g = 1.f;       // unit backward: every per-row gradient below is linear in g (see k_bwd_rows_pair)
            dpre = g;
        }
        float4 dxh[RV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i);
            dxh[i] = f4_zero();
            if (c < H2 && ok) {
                const float4 dz = make_float4(h2v[rr][i].x > 0.f ? dpre * w3c[i].x : 0.f, h2v[rr][i].y > 0.f ? dpre * w3c[i].y : 0.f,
                                              h2v[rr][i].z > 0.f ? dpre * w3c[i].z : 0.f, h2v[rr][i].w > 0.f ? dpre * w3c[i].w : 0.f);
                *reinterpret_cast<float4 *>(o.dz + (size_t)row * H2 + c) = dz;
                dxh[i] = make_float4(dz.x * g2c[i].x, dz.y * g2c[i].y, dz.z * g2c[i].z, dz.w * g2c[i].w);
                s1 += f4_sum(dxh[i]);
                s2 = fmaf(dxh[i].x, xh[rr][i].x, s2); s2 = fmaf(dxh[i].y, xh[rr][i].y, s2);
                s2 = fmaf(dxh[i].z, xh[rr][i].z, s2); s2 = fmaf(dxh[i].w, xh[rr][i].w, s2);
            }
        }
        s1 = wave_sum64(s1) * (1.f / H2);
        s2 = wave_sum64(s2) * (1.f / H2);
        float4 vv[RV];
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int c = rv_col(lane, i);
            float4 v = f4_zero();
            if (c < H2 && ok) {
                v = make_float4(rs[rr] * (dxh[i].x - s1 - xh[rr][i].x * s2), rs[rr] * (dxh[i].y - s1 - xh[rr][i].y * s2),
                                rs[rr] * (dxh[i].z - s1 - xh[rr][i].z * s2), rs[rr] * (dxh[i].w - s1 - xh[rr][i].w * s2));
                *reinterpret_cast<float4 *>(o.dx2 + (size_t)row * H2 + c) = v;
            }
            vv[i] = v;
            if (!img && c < DS) *reinterpret_cast<float4 *>(&dx2_s[lr * DS + c]) = v;     // zero in the K padding (columns 300..307)
        }
        if (img) {
            // the row as two f16 planes, scaled by the power of two that puts its largest entry in [2^12, 2^13): gradients
            // are far below f16's normal range as they come
            float mx = 0.f;
#pragma unroll
            for (int i = 0; i < RV; ++i)
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(vv[i].x), fabsf(vv[i].y))), fmaxf(fabsf(vv[i].z), fabsf(vv[i].w)));
            mx = wave_max64(mx);
            const int e = (__builtin_bit_cast(int, mx) >> 23) & 0xff;
            const int se = min(max(266 - e, 1), 253);
            const float sc = __builtin_bit_cast(float, se << 23), inv = __builtin_bit_cast(float, (254 - se) << 23);
#pragma unroll
            for (int i = 0; i < RV; ++i) {                  // c < 320: every plane entry, zeros beyond column 299
                const int c = rv_col(lane, i);
                if (c < N2P) {
                    uint2 ph, pm;
                    split4(make_float4(vv[i].x * sc, vv[i].y * sc, vv[i].z * sc, vv[i].w * sc), ph, pm);
                    *reinterpret_cast<uint2 *>(dxp_s + lr * DSH + c) = ph;
                    *reinterpret_cast<uint2 *>(dxp_s + TR * DSH + lr * DSH + c) = pm;
                }
            }
            if (lane == 0) rsc_s[lr] = inv * (1.f / SWL);
        }
        if (lane == 0 && ok) o.dpre[row] = dpre;
    }
    lds_barrier();
    STAMP(9);
    // ---- phase B: dH1 [16,400] = dX2 [16,304] * W2 [304,400].  Wave w (< 7) owns the 64-column group w; within the
    // group, output tile t holds columns c0 + 4*(l&15) + t (one float4 of a W2 row feeds the 4 tiles).
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int c0 = wave * 64 + 4 * l15;
    const bool gok = wave < NG && c0 < H1;
    // what phase C needs from memory is requested now, so that it arrives behind phase B's MFMAs (and, as in phase A,
    // before any store of phase C)
    float4 hv[4], xv[4];
    // (unconditional, from clamped rows / columns, as in phase A: phase C uses them under `gok && row < n` only)
    const int c0c = gok ? c0 : 0;
    const float4 gm = f4_ldu(W.g1 + c0c);
    float rs1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rowc = min(row0 + l4 * 4 + r, n - 1);
        hv[r] = f4_ldu(sv.h1 + (size_t)rowc * H1 + c0c);
        xv[r] = f4_ldu(sv.xh1 + (size_t)rowc * H1 + c0c);
        rs1[r] = sv.rstd1[rowc];
    }
    if (img && wave < NG) {
        // the image's backward half: this wave's four tiles (= its 64-column group) in k32 steps over n; A = the dX2 planes
        const _Float16 *ah = dxp_s + l15 * DSH + 8 * l4;
        const _Float16 *bh = W.img + 2 * IMG_FWD + ((size_t)wave * 4 * BW_STEPS * 64 + lane) * 8;
        constexpr int GU = 2, NGRP = BW_STEPS / GU;
        f16x8 bcur[GU][4][2], bnxt[GU][4][2];
        auto load_group = [&](const int g, f16x8 (&dst)[GU][4][2]) {
#pragma unroll
            for (int u = 0; u < GU; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dst[u][t][0] = *reinterpret_cast<const f16x8 *>(bh + (size_t)(t * BW_STEPS + g * GU + u) * 512);
                    dst[u][t][1] = *reinterpret_cast<const f16x8 *>(bh + IMG_T + (size_t)(t * BW_STEPS + g * GU + u) * 512);
                }
        };
        static_assert(GU == GU_B, "the prefetched group is group 0");
        // two fragment sets that swap roles (no copies: see the forward's layer 2)
        auto products = [&](const int g, const f16x8 (&b)[GU][4][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int c = g * GU + u;
                const f16x8 a_h = *reinterpret_cast<const f16x8 *>(ah + 32 * c);
                const f16x8 a_m = *reinterpret_cast<const f16x8 *>(ah + TR * DSH + 32 * c);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma_h(a_m, b[u][t][0], acc[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma_h(a_h, b[u][t][1], acc[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma_h(a_h, b[u][t][0], acc[t]);
            }
        };
        auto step = [&](const int g, const f16x8 (&use)[GU][4][2], f16x8 (&fill)[GU][4][2]) __attribute__((always_inline)) {
            if (g + 1 < NGRP) load_group(g + 1, fill);
            __builtin_amdgcn_sched_barrier(0);
            products(g, use);
            __builtin_amdgcn_sched_barrier(0);
        };
        (void)bcur;
#pragma unroll
        for (int g = 0; g < NGRP; g += 2) {
            if (g == 0) step(0, bpre, bnxt);
            else step(g, bcur, bnxt);
            if (g + 1 < NGRP) step(g + 1, bnxt, bcur);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float back = rsc_s[l4 * 4 + r];          // 1 / (the row's scale * 64)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][r] *= back;
        }
    } else if (wave < NG) {
        const float *arow = dx2_s + l15 * DS + 4 * l4;
        const float *wcol = W.w2 + (gok ? c0 : 0);
        // software pipeline as in the forward's layer 2: the fc2 rows of the NEXT group of 3 k16 steps are requested before
        // the 48 MFMAs of the current group issue (19 steps = 6 groups of 3 + 1)
        constexpr int GU = 3, NSTEP = H2K / 16, NGRP = (NSTEP + GU - 1) / GU;
        float4 bcur[GU][4], bnxt[GU][4];
        auto load_group = [&](const int g, float4 (&dst)[GU][4]) {
#pragma unroll
            for (int u = 0; u < GU; ++u)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int c = g * GU + u, j = 16 * c + 4 * l4 + ks;          // row of fc2 = the k of this product
                    dst[u][ks] = (gok && c < NSTEP && j < H2) ? *reinterpret_cast<const float4 *>(wcol + (size_t)j * H1)
                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        };
        load_group(0, bcur);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            if (g + 1 < NGRP) load_group(g + 1, bnxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int c = g * GU + u;
                if (c < NSTEP) {
                    const float4 av4 = *reinterpret_cast<const float4 *>(arow + 16 * c);
                    const float a[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bcur[u][ks].x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bcur[u][ks].y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bcur[u][ks].z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bcur[u][ks].w, acc[3], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GU; ++u)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bcur[u][ks] = bnxt[u][ks];
        }
    }
    STAMP(10);
    // ---- phase C: ReLU and LayerNorm1 backward; accumulator [t][r] is row l4*4+r, column wave*64 + 4*l15 + t
    float xh1[4][4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + l4 * 4 + r;
        const float4 h = hv[r], x = xv[r];
        const bool ok = gok && row < n;
        const float hh[4] = {h.x, h.y, h.z, h.w}, xx[4] = {x.x, x.y, x.z, x.w}, gg[4] = {gm.x, gm.y, gm.z, gm.w};
        float dy[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dy[t] = (ok && hh[t] > 0.f) ? acc[t][r] : 0.f;
            xh1[t][r] = xx[t];
            acc[t][r] = dy[t] * gg[t];                       // d(x-hat)
            s1[r] += acc[t][r];
            s2[r] = fmaf(acc[t][r], xx[t], s2[r]);
        }
        if (ok) *reinterpret_cast<float4 *>(o.dy1 + (size_t)row * H1 + c0) = make_float4(dy[0], dy[1], dy[2], dy[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = row_sum16(s1[r]); s2[r] = row_sum16(s2[r]); }
    cross_wave_sum2(red, wave, l4, l15, s1, s2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + l4 * 4 + r;
        if (gok && row < n) {
            const float rs = rs1[r], m1 = s1[r] * (1.f / H1), m2 = s2[r] * (1.f / H1);
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = rs * (acc[t][r] - m1 - xh1[t][r] * m2);
            *reinterpret_cast<float4 *>(o.dx1 + (size_t)row * H1 + c0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    STAMP(11);
}

template <bool CRITIC>
__global__ __launch_bounds__(64 * NW) void k_bwd_rows(const int n, const int mode, const float scale,
                                                      const float *__restrict__ d_out, const float *__restrict__ out,
                                                      const float *__restrict__ y, const float *__restrict__ aux,
                                                      const Weights W, const Saved sv, const BwdOut o, const TdIn td) {
    __shared__ __attribute__((aligned(16))) float dx2_s[DXS_FLOATS];
    __shared__ float red[2 * NW * TR];
    __shared__ float rsc_s[TR];
    bwd_rows_body<CRITIC>(n, mode, scale, d_out, out, y, aux, W, sv, o, td, dx2_s, red, rsc_s, blockIdx.x * TR);
}

// The critic's per-row backward (TD prologue, mode 1) and the ACTOR's unit backward (mode 3) in one launch, on different
// workgroups.  The actor's per-row gradients are linear in the row's d(loss)/d(pre-tanh) = -(1/B) dQ/da (1 - mu^2), and dQ/da
// needs the UPDATED critic (DDPG_agent.py:100-103) -- but everything else of the actor's backward (ReLU masks, both
// LayerNorm backwards, dH1 = dX2 * W2) only needs what the forward saved.  So that part runs HERE, beside the critic's
// backward, for a unit gradient, and k_bwd_weights multiplies row b by the real number once the critic has been updated and
// dQ/da is known: the actor's backward is off the chain's critical path.
// Optional rider: the pack of a vector step's policy image (csrc/ttnet_pack.h) on IMAGE_WGS further workgroups of this launch.
struct ImageJob {
    ttnet::Weights W;
    unsigned char *ws, *ws_alt;
    ttnet::RingCursor cur;
    int on;
};
constexpr int IMAGE_WGS = (ttnet::PACK_THREADS + 64 * NW - 1) / (64 * NW);

__global__ __launch_bounds__(64 * NW) void k_bwd_rows_pair(const int n, const float scale_c, const float *__restrict__ q_out,
                                                           const Weights Wc, const Saved sv_c, const BwdOut o_c, const TdIn td,
                                                           const float *__restrict__ mu_out, const Weights Wa, const Saved sv_a,
                                                           const BwdOut o_a, const ImageJob img) {
    __shared__ __attribute__((aligned(16))) float dx2_s[DXS_FLOATS];
    __shared__ float red[2 * NW * TR];
    __shared__ float rsc_s[TR];
    kernarg_warm<16 + 2 * ((int)sizeof(Weights) + (int)sizeof(Saved) + (int)sizeof(BwdOut)) + (int)sizeof(TdIn) + 8>();
    const int nb = (n + TR - 1) / TR;
    if ((int)blockIdx.x == 2 * nb) {         // the extra workgroup: counters + bias corrections (nothing in this launch reads them)
        if (threadIdx.x == 0) clock_tick(td);
        return;
    }
    if ((int)blockIdx.x > 2 * nb) {          // the image's workgroups (they read a SNAPSHOT of the step number: see ImageJob)
        ttnet::split_pack_body(img.W, false, img.ws, img.ws_alt, nullptr, img.cur,
                               ((int)blockIdx.x - 2 * nb - 1) * (64 * NW) + (int)threadIdx.x);
        ttnet::publish_image(img.cur, IMAGE_WGS);
        return;
    }
    KBEGIN(1);
    if ((int)blockIdx.x < nb) {
        bwd_rows_body<true>(n, 1, scale_c, nullptr, q_out, nullptr, nullptr, Wc, sv_c, o_c, td, dx2_s, red, rsc_s, blockIdx.x * TR);
    } else {
        const TdIn none{};
        bwd_rows_body<false>(n, 3, 1.f, nullptr, mu_out, nullptr, nullptr, Wa, sv_a, o_a, none, dx2_s, red, rsc_s,
                             ((int)blockIdx.x - nb) * TR);
    }
    KEND(1);
}

// ------------------------------------------------------------------------------------------------------
// weight gradients.  Workgroup roles by blockIdx; K = batch in permuted k16 steps, SPLIT over the 4 waves of the
// workgroup (each wave takes a quarter of the batch rows, all its loads are independent and issued together), the
// four partial tiles are then added through LDS in a fixed order (deterministic, no atomics):
//   [0, NU2)            dW2 [300,400] = dX2^T * H1: one (16 rows j) x (64-column group) block per workgroup; 19 x 7
//   [NU2, NU2+NU1)      dW1 [400,23]  = dX1^T * S : 16 rows j x 2 column tiles per workgroup; 25
//   then                column sums (db2, dg2, dbe2, db1, dg1, dbe1, dw3, db3, critic: dwa, dba): a workgroup sums 64
//                       columns, its 4 waves a quarter of the rows each
// Optional optimizer step inside k_bwd_weights: the workgroup that finishes a gradient element owns it (K = batch is
// never split across workgroups), so it can apply torch.optim.Adam and the soft target update to that element right
// away -- the arithmetic of k_adam_soft below, one launch and one pass over the gradient less per network.  Not used
// when the gradients are all-reduced across ranks first.  Tensor order: w1 b1 g1 be1 w2 b2 g2 be2 w3 b3 wa ba.
struct AdamFused {
    float *p[12], *m[12], *v[12], *tgt[12];
    const long long *step_dev;
    float lr, beta1, beta2, eps, weight_decay, tau;
    int on;
    _Float16 *img_p, *img_t;      // fc2 images of the network / its target that this step keeps current (or nullptr)
    const float *bias_corr;       // clock_tick()'s {step, beta1, beta2, 1 - beta1^t, 1 - beta2^t} or nullptr
};

// Adam's bias corrections for step t: from clock_tick()'s buffer when it holds exactly this step and these betas (c = its five
// floats, loaded by the caller with its other loads), else evaluated here
__device__ __forceinline__ void adam_bias_corrections(const float beta1, const float beta2, const long long t, const bool have,
                                                      const float (&c)[5], float &bc1, float &bc2) {
    if (have && __float_as_int(c[0]) == (int)t && c[1] == beta1 && c[2] == beta2) {
        bc1 = c[3]; bc2 = c[4];
    } else {
        bc1 = (float)(1.0 - pow((double)beta1, (double)t));
        bc2 = (float)(1.0 - pow((double)beta2, (double)t));
    }
}

struct AdamElem { float p, m, v, tg; };

__device__ __forceinline__ AdamElem adam_load(const AdamFused &A, const int t, const size_t i) {
    return AdamElem{A.p[t][i], A.m[t][i], A.v[t][i], A.tgt[t] ? A.tgt[t][i] : 0.f};
}

// k_adam_soft's arithmetic on one element already loaded; returns {new parameter, new target}
__device__ __forceinline__ float2 adam_finish(const AdamFused &A, const int t, const size_t i, const float grad, AdamElem e,
                                              const float bc1, const float sqrt_bc2) {
    const float g = fmaf(A.weight_decay, e.p, grad);
    const float m = fmaf(A.beta1, e.m, (1.f - A.beta1) * g);
    const float v = fmaf(A.beta2, e.v, (1.f - A.beta2) * g * g);
    st_out(&A.m[t][i], m);
    st_out(&A.v[t][i], v);
    const float denom = sqrtf(v) / sqrt_bc2 + A.eps;
    const float p = e.p - (A.lr / bc1) * (m / denom);
    st_out(&A.p[t][i], p);
    float tg = e.tg;
    if (A.tgt[t]) {
        tg = fmaf(A.tau, p - e.tg, e.tg);
        st_out(&A.tgt[t][i], tg);
    }
    return make_float2(p, tg);
}

// the same on one tensor's four arrays picked beforehand (a quantity chosen at run time: indexing the tables of AdamFused
// with a run-time tensor number keeps all 48 pointers live in scalar registers)
struct AdamPtrs { float *p, *m, *v, *tgt; };
__device__ __forceinline__ AdamElem adam_load(const AdamPtrs &q, const size_t i) {
    return AdamElem{q.p[i], q.m[i], q.v[i], q.tgt ? q.tgt[i] : 0.f};
}
__device__ __forceinline__ void adam_finish(const AdamFused &A, const AdamPtrs &q, const size_t i, const float grad, AdamElem e,
                                            const float bc1, const float sqrt_bc2) {
    const float g = fmaf(A.weight_decay, e.p, grad);
    const float m = fmaf(A.beta1, e.m, (1.f - A.beta1) * g);
    const float v = fmaf(A.beta2, e.v, (1.f - A.beta2) * g * g);
    st_out(&q.m[i], m);
    st_out(&q.v[i], v);
    const float denom = sqrtf(v) / sqrt_bc2 + A.eps;
    const float p = e.p - (A.lr / bc1) * (m / denom);
    st_out(&q.p[i], p);
    if (q.tgt) st_out(&q.tgt[i], fmaf(A.tau, p - e.tg, e.tg));
}

// Optional per-row factor of k_bwd_weights' inputs: row b of dpre / dz / dx2 / dy1 / dx1 (a unit backward, mode 3) counts
// f(b) = scale * dq_da[b] * (1 - mu[b]^2) times: the actor's d(loss)/d(pre-tanh) for loss = -mean Q(s, mu(s)).
struct RowScale {
    const float *__restrict__ dq_da, *__restrict__ mu;
    float scale;
};
constexpr int MAXB = 1024;     // rows whose factors fit the LDS table of k_bwd_weights<true> (tt_mlp_backward_weights checks)

struct Grads {
    float *__restrict__ w1, *__restrict__ b1, *__restrict__ g1, *__restrict__ be1, *__restrict__ w2, *__restrict__ b2,
        *__restrict__ g2, *__restrict__ be2, *__restrict__ w3, *__restrict__ b3, *__restrict__ wa, *__restrict__ ba;
};
constexpr int JT2 = (H2 + 15) / 16;                 // 19 row tiles of dW2
constexpr int NU2 = JT2 * NG;                       // 133 workgroups
constexpr int NU1 = H1 / 16;                        // 25 workgroups
constexpr int NCAT = 10;                            // db2 dg2 dbe2 db1 dg1 dbe1 dw3 db3 dwa dba
constexpr int SUMB_ACTOR = 3 * 5 + 3 * 7 + 5 + 1;   // 64-column chunks per quantity: 300 -> 5, 400 -> 7, 1 -> 1
constexpr int SUMB_CRITIC = SUMB_ACTOR + 2 * 5;

// Order of events in every workgroup of k_bwd_weights: (1) EVERYTHING it will read is requested at once -- the operands of its
// first 64 batch rows per wave (the whole batch at 256 rows), the optimizer state of the elements it will finish, the step count;
// (2) [ROWSCALE] the rows' factors are formed into LDS while those loads fly; (3) products / sums; (4) bias corrections
// (two f64 pow() -- behind the loads, not in front of them), Adam, soft update, image patch.  Round 2's order (factor table ->
// barrier -> loop of 16-row steps, each with its own load round trip -> optimizer state loads) cost five dependent memory
// round trips per workgroup; this one has one.
constexpr int KCH = 4;                              // k16 steps per chunk of a wave's batch rows
// (register budget: <= 168, three workgroups per CU -- beside the policy's grid only ~85 CUs are free for the ~205 of this launch)
// Hand-over of dQ/da INSIDE one launch (k_actor_tail below): the row workgroups of Q(s, mu(s)) publish, the weight-gradient
// workgroups of the same grid -- everything else they need already requested -- wait.  Per ROW one 8-byte word {learn step, dQ/da}
// written by ONE agent-scope atomic store and read by one agent-scope atomic load: a reader that sees the step has the value, with
// no ordering between two locations to rely on and no cache maintenance on either side.  Per row WORKGROUP one hint word (the
// learn step, stored after the rows' words): a consumer polls the 16 hints with one wave before its 256 threads look at the rows,
// so that 200 waiting workgroups do not hammer 256 words; the hints prove nothing, the rows' words do.
struct TailSync {
    int *hints;                          // [64] device ints (nullptr: no hand-over, dQ/da is complete when the launch starts)
    unsigned long long *rows;            // [n] {step << 32 | float bits of dQ/da}
    int producers;
    int *gave_up_host;                   // one int of pinned host memory: set (system scope) by a consumer that stopped waiting
};
__device__ __forceinline__ void tail_wait_hints(const TailSync &ts, const int epoch) {      // the first wave of the workgroup calls this
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const int v = lane < ts.producers ? __hip_atomic_load(ts.hints + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        if (__all(v == epoch)) break;
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > ttnet::TT_IMAGE_WAIT_TICKS) break;      // (the rows' words decide, below)
    }
}
// dQ/da of row b for learn step `epoch` (bounded: never hang -- mark, host-visible, and go on; the caller raises)
__device__ __forceinline__ float tail_row(const TailSync &ts, const int b, const int epoch) {
    unsigned long long w = __hip_atomic_load(ts.rows + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(w >> 32) != epoch) {
        const unsigned long long t0 = wall_clock64();
        do {
            __builtin_amdgcn_s_sleep(8);
            w = __hip_atomic_load(ts.rows + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wall_clock64() - t0 > ttnet::TT_IMAGE_WAIT_TICKS) {
                if (ts.gave_up_host) __hip_atomic_store(ts.gave_up_host, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        } while ((int)(w >> 32) != epoch);
    }
    return __uint_as_float((unsigned)w);
}

template <bool ROWSCALE, bool TAIL>
__device__ __forceinline__ void bwd_weights_body(const int blk, const int n, const int critic, const float *__restrict__ obs,
                                                 const float *__restrict__ action, const Saved &sv, const BwdOut &d, const Grads &G,
                                                 const AdamFused &A, const RowScale &RS, float (&part)[4][4][256],
                                                 float *__restrict__ f_s, const TailSync &ts, const long long tail_epoch,
                                                 _Float16 *__restrict__ stage_s) {
    // stage_s: 4 x 1024 halves of LDS (8 KB) of the workgroup's own: the forward-image pieces of a dW2 workgroup's patch
    auto fill_factors = [&]() __attribute__((always_inline)) {       // every thread of the workgroup calls this once
        if (ROWSCALE) {
            if (TAIL && ts.hints) {
                // everything else this workgroup reads is in flight by now: wait for the producers' rows of THIS learn step
                if (threadIdx.x < 64) tail_wait_hints(ts, (int)tail_epoch);
                lds_barrier();
                for (int b = threadIdx.x; b < n; b += 256) {
                    const float m = RS.mu[b];
                    f_s[b] = RS.scale * tail_row(ts, b, (int)tail_epoch) * (1.f - m * m);
                }
            } else {
                for (int b = threadIdx.x; b < n; b += 256) {
                    const float m = RS.mu[b];
                    f_s[b] = RS.scale * RS.dq_da[b] * (1.f - m * m);
                }
            }
            lds_barrier();
        }
    };
    auto row_factor = [&](const int b) -> float { return ROWSCALE ? f_s[b] : 1.f; };
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    KBEGIN(ROWSCALE ? 4 : 2);
    STAMPB(12, 0); STAMPB(14, NU2); STAMPB(5, NU2 + NU1);
#ifdef TT_STAMPS
    if (threadIdx.x == 0) g_blk[blockIdx.x][0] = wall_clock64();
#endif
    WST(8);
    // The step count and Adam's bias corrections are needed at the very end, but wherever they are read the compiler turns the two
    // wave-uniform reads into SCALAR loads through pointers that are themselves kernel arguments and hoists them to the top:
    // kernarg -> pointer -> value, two dependent round trips of the scalar cache behind ONE counter (lgkmcnt), in front of the
    // first operand load of every workgroup (~0.7 us of the 1.5 us "entry -> operand loads issued" of round 3's stamps).  With
    // the pointers laundered into vector registers they are ordinary vector loads: asynchronous, first in the queue.
    long long step_count = 0;
    float bcc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (A.on) {
        const long long *sp = A.step_dev;
        asm volatile("" : "+v"(sp));
        step_count = *sp;
        if (A.bias_corr) {
            const float *bp = A.bias_corr;
            asm volatile("" : "+v"(bp));
            const float4 b4 = *reinterpret_cast<const float4 *>(bp);
            bcc[0] = b4.x; bcc[1] = b4.y; bcc[2] = b4.z; bcc[3] = b4.w; bcc[4] = bp[4];
        }
    }
    auto bias_corrections = [&](float &bc1, float &sqrt_bc2) __attribute__((always_inline)) {
        bc1 = 1.f; sqrt_bc2 = 1.f;
        if (A.on) {
            float bc2;
            adam_bias_corrections(A.beta1, A.beta2, step_count, A.bias_corr != nullptr, bcc, bc1, bc2);
            sqrt_bc2 = sqrtf(bc2);
        }
    };
    const int rows_w = (((n + 3) / 4) + 15) / 16 * 16;                  // batch rows per wave, whole k16 steps
    const int b_lo = wave * rows_w, b_hi = min(n, b_lo + rows_w);
    if (blk < NU2) {
        const int jt = blk / NG, grp = blk - jt * NG;
        const int j = jt * 16 + l15, c0 = grp * 64 + 4 * l15;
        const bool jok = j < H2, cok = c0 < H1;
        // Who finishes which of the block's 16 x 64 outputs: wave w rows 4w .. 4w+3, lane = column (round 4).  The optimizer state
        // and the updated weights then move as whole 256-byte rows per instruction (with "wave = column mod 4" every instruction
        // touched four rows at a quarter of each line), and the eight columns of a 16-byte image piece sit in eight lanes of ONE
        // wave: the patch needs no workgroup barrier.
        const int col = grp * 64 + lane;
        const bool own = col < H1;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float av[KCH][4];
        float4 bv[KCH][4];
        // unconditional loads from clamped (always valid) addresses, zeroed afterwards: a guarded load costs a saved exec
        // mask each, and all of them are in flight together
        {
            const int jc = jok ? j : 0, cc = cok ? c0 : 0;
#pragma unroll
            for (int it = 0; it < KCH; ++it)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int b = b_lo + 16 * it + 4 * l4 + ks, bc = min(b, n - 1);              // permuted k order
                    av[it][ks] = d.dx2[(size_t)bc * H2 + jc];                                      // A[i = j][k = b]
                    bv[it][ks] = *reinterpret_cast<const float4 *>(sv.h1 + (size_t)bc * H1 + cc);
                }
#pragma unroll
            for (int it = 0; it < KCH; ++it)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int b = b_lo + 16 * it + 4 * l4 + ks;
                    if (!(b < b_hi && jok)) av[it][ks] = 0.f;
                    if (!(b < b_hi && cok)) bv[it][ks] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        WST(9);
        // the four elements' optimizer state: requested with the operands (the updates below also store through pointers that
        // may alias a later load)
        AdamElem el[4] = {};
        if (A.on && own) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jr = jt * 16 + wave * 4 + r;
                if (jr < H2) el[r] = adam_load(A, 4, (size_t)jr * H1 + col);
            }
        }
        WST(0);
        fill_factors();
        WST(1);
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
            float a[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int b = b_lo + 16 * it + 4 * l4 + ks;
                a[ks] = ROWSCALE ? (b < b_hi ? av[it][ks] * row_factor(b) : 0.f) : av[it][ks];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bv[it][ks].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bv[it][ks].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bv[it][ks].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bv[it][ks].w, acc[3], 0, 0, 0);
            }
        }
        for (int b0 = b_lo + 16 * KCH; b0 < b_hi; b0 += 16) {          // (batches beyond 256 rows: one k16 step at a time)
            float a[4];
            float4 bb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int b = b0 + 4 * l4 + ks;
                a[ks] = (b < b_hi && jok) ? d.dx2[(size_t)b * H2 + j] * row_factor(b) : 0.f;
                bb[ks] = (b < b_hi && cok) ? *reinterpret_cast<const float4 *>(sv.h1 + (size_t)b * H1 + c0)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bb[ks].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bb[ks].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bb[ks].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bb[ks].w, acc[3], 0, 0, 0);
            }
        }
        WST(2);
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(&part[wave][t][lane * 4]) = acc[t];
        float bc1, sqrt_bc2;
        bias_corrections(bc1, sqrt_bc2);
        WST(3);
        lds_barrier();
        WST(4);
        // element (row 4 wave + r, column lane) of the block: tile t = lane & 3 of the MFMA layout, held there by lane
        // wave * 16 + (lane >> 2), register r; the four K-quarters summed in a fixed order
        const int src = (wave * 16 + (lane >> 2)) * 4, tq = lane & 3;
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(&part[0][tq][src]);
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(&part[1][tq][src]);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(&part[2][tq][src]);
        const f32x4 p3 = *reinterpret_cast<const f32x4 *>(&part[3][tq][src]);
        float pnew[4] = {0.f, 0.f, 0.f, 0.f}, tnew[4] = {0.f, 0.f, 0.f, 0.f};      // updated parameter / target (0 = padding)
        if (own) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jr = jt * 16 + wave * 4 + r;
                if (jr < H2) {
                    const float g = ((p0[r] + p1[r]) + p2[r]) + p3[r];
                    st_out(&G.w2[(size_t)jr * H1 + col], g);
                    if (A.on) {
                        const float2 pt = adam_finish(A, 4, (size_t)jr * H1 + col, g, el[r], bc1, sqrt_bc2);
                        pnew[r] = pt.x; tnew[r] = pt.y;
                    }
                }
            }
        }
        WST(5);
        if (A.on && (A.img_p || A.img_t)) {
            // The block's 16 x 64 patch of fc2 in the images the learn() kernels read instead of w2 (fragment order, see IMG_FWD
            // above): two consecutive 1 KB fragments of each forward plane (rows = tile jt, k32 steps 2 grp, 2 grp + 1) -- a
            // 16-byte piece is one row's eight consecutive columns: eight lanes of this wave, put together in the wave's share of
            // stage_s (LDS operations of one wave complete in order: no barrier) -- and half a fragment of each of the group's four
            // backward tiles, where a lane's four rows ARE four consecutive halves: 8-byte stores straight from registers.
            const int kq = lane;                                          // column inside the group
            uint32_t bh[2] = {0u, 0u}, bm[2] = {0u, 0u};                  // this lane's four rows of the backward planes (h, m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = wave * 4 + r;                              // row inside the tile
                const float sp = pnew[r] * SWL, st = tnew[r] * SWL;
                const _Float16 ph = (_Float16)sp, pm = (_Float16)(sp - (float)ph), th = (_Float16)st, tm = (_Float16)(st - (float)th);
                const int f = (kq >> 5) * 512 + (((kq >> 3) & 3) * 16 + lr) * 8 + (kq & 7);
                stage_s[f] = ph; stage_s[1024 + f] = pm; stage_s[2048 + f] = th; stage_s[3072 + f] = tm;
                bh[r >> 1] |= (uint32_t)__builtin_bit_cast(unsigned short, ph) << (16 * (r & 1));
                bm[r >> 1] |= (uint32_t)__builtin_bit_cast(unsigned short, pm) << (16 * (r & 1));
            }
            if (A.img_p) {
                // backward planes: piece (kq & 3) * 32 + (wave >> 1) * 16 + (kq >> 2) of the staging order, halves (wave & 1) * 4 .. + 3
                const size_t o = 2 * IMG_FWD + ((size_t)((grp * 4 + (kq & 3)) * BW_STEPS + (jt >> 1)) * 64 + (jt & 1) * 32 + (wave >> 1) * 16 + (kq >> 2)) * 8 +
                                 (wave & 1) * 4;
                *reinterpret_cast<uint2 *>(A.img_p + o) = make_uint2(bh[0], bh[1]);
                *reinterpret_cast<uint2 *>(A.img_p + o + IMG_T) = make_uint2(bm[0], bm[1]);
            }
            __builtin_amdgcn_wave_barrier();
            const size_t fbase = (size_t)(jt * FW_STEPS + 2 * grp) * 512;
            const int fcount = grp < NG - 1 ? 128 : 64;                   // the last group has one k32 step (columns 384..415)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // this wave's 128 pieces: array (p.h, p.m, t.h, t.m) x k32 half x k8 group x its four rows
                const int pc = lane + 64 * i, arr = pc >> 5, w = ((pc >> 4) & 1) * 64 + ((pc >> 2) & 3) * 16 + wave * 4 + (pc & 3);
                const uint4 v = *reinterpret_cast<const uint4 *>(&stage_s[arr * 1024 + w * 8]);
                if (w < fcount) {
                    _Float16 *img = arr < 2 ? A.img_p : (A.tgt[4] ? A.img_t : nullptr);
                    if (img) st_out16(img + ((arr & 1) ? IMG_FWD : 0) + fbase + w * 8, v);
                }
            }
        }
        WST(6);
        STAMPB(13, 0);
#ifdef TT_STAMPS
        __syncthreads(); if (threadIdx.x == 0) g_blk[blockIdx.x][1] = wall_clock64();
#endif
        WST(7);
    } else if (blk < NU2 + NU1) {
        const int jt = blk - NU2, j = jt * 16 + l15;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        float av[KCH][4], b0v[KCH][4], b1v[KCH][4];
#pragma unroll
        for (int it = 0; it < KCH; ++it)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int b = b_lo + 16 * it + 4 * l4 + ks, bc = min(b, n - 1);
                av[it][ks] = d.dx1[(size_t)bc * H1 + j];
                b0v[it][ks] = obs[(size_t)bc * IN + l15];                                            // columns 0..15
                b1v[it][ks] = obs[(size_t)bc * IN + (16 + l15 < IN ? 16 + l15 : 0)];                 // 16..22
            }
#pragma unroll
        for (int it = 0; it < KCH; ++it)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int b = b_lo + 16 * it + 4 * l4 + ks;
                if (!(b < b_hi)) { av[it][ks] = 0.f; b0v[it][ks] = 0.f; }
                if (!(b < b_hi && 16 + l15 < IN)) b1v[it][ks] = 0.f;
            }
        const int col = wave * 16 + l15;
        const bool own = wave < 2 && col < IN;
        AdamElem el[4] = {};
        if (A.on && own) {
#pragma unroll
            for (int r = 0; r < 4; ++r) el[r] = adam_load(A, 0, (size_t)(jt * 16 + l4 * 4 + r) * IN + col);
        }
        fill_factors();
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int b = b_lo + 16 * it + 4 * l4 + ks;
                const float a = ROWSCALE ? (b < b_hi ? av[it][ks] * row_factor(b) : 0.f) : av[it][ks];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0v[it][ks], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1v[it][ks], acc[1], 0, 0, 0);
            }
        }
        for (int b0 = b_lo + 16 * KCH; b0 < b_hi; b0 += 16) {          // (batches beyond 256 rows)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int b = b0 + 4 * l4 + ks;
                const float a = b < b_hi ? d.dx1[(size_t)b * H1 + j] * row_factor(b) : 0.f;
                const float x0 = b < b_hi ? obs[(size_t)b * IN + l15] : 0.f;
                const float x1 = (b < b_hi && 16 + l15 < IN) ? obs[(size_t)b * IN + 16 + l15] : 0.f;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x1, acc[1], 0, 0, 0);
            }
        }
        *reinterpret_cast<f32x4 *>(&part[wave][0][lane * 4]) = acc[0];
        *reinterpret_cast<f32x4 *>(&part[wave][1][lane * 4]) = acc[1];
        float bc1, sqrt_bc2;
        bias_corrections(bc1, sqrt_bc2);
        lds_barrier();
        if (own) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = ((part[0][wave][lane * 4 + r] + part[1][wave][lane * 4 + r]) + part[2][wave][lane * 4 + r]) +
                                part[3][wave][lane * 4 + r];
                st_out(&G.w1[(size_t)(jt * 16 + l4 * 4 + r) * IN + col], g);
                if (A.on) adam_finish(A, 0, (size_t)(jt * 16 + l4 * 4 + r) * IN + col, g, el[r], bc1, sqrt_bc2);
            }
        }
        STAMPB(15, NU2);
#ifdef TT_STAMPS
        __syncthreads(); if (threadIdx.x == 0) g_blk[blockIdx.x][1] = wall_clock64();
#endif
    } else {
        // column sums.  Every workgroup handles 64 columns of ONE quantity (a wave straddling two quantities would run
        // both row loops one after the other): [db2 dg2 dbe2 | db1 dg1 dbe1 | dw3 | db3 | dwa dba] in 64-column chunks
        int cb = blk - NU2 - NU1;
        int cat = -1, base = 0;
#pragma unroll
        for (int k = 0; k < NCAT; ++k) {
            const int width = k < 3 ? H2 : (k < 6 ? H1 : (k == 7 ? 1 : H2));
            const int nb = (width + 63) / 64;
            if (cat < 0) {
                if (cb < nb) { cat = k; base = width; }
                else cb -= nb;
            }
        }
        const int c = cb * 64 + lane;
        const bool valid = cat >= 0 && c < base;
        // every quantity is sum_b A[b*sa + ca] (* B[b*sb + cb]); the operands are picked ONCE per workgroup so that the
        // row loops below are straight-line: the loads of 64 rows in flight at once, then one add chain in row order.
        // (A per-term switch on the quantity serialises the loads: one L2/HBM round trip per row.)
        const float *Ap = nullptr, *Bp = nullptr;
        int sa = 0, sb = 0, ca = 0, cbb = 0;
        switch (cat) {
            case 0: Ap = d.dx2; sa = H2; ca = c; break;                                   // db2
            case 1: Ap = d.dz; sa = H2; ca = c; Bp = sv.xh2; sb = H2; cbb = c; break;     // dg2 = sum dz * xh2
            case 2: Ap = d.dz; sa = H2; ca = c; break;                                    // dbe2
            case 3: Ap = d.dx1; sa = H1; ca = c; break;                                   // db1
            case 4: Ap = d.dy1; sa = H1; ca = c; Bp = sv.xh1; sb = H1; cbb = c; break;    // dg1 = sum dy1 * xh1
            case 5: Ap = d.dy1; sa = H1; ca = c; break;                                   // dbe1
            case 6: Ap = d.dpre; sa = 1; ca = 0; Bp = sv.h2; sb = H2; cbb = c; break;     // dw3 = sum dpre * h2
            case 7: Ap = d.dpre; sa = 1; ca = 0; break;                                   // db3
            case 8: Ap = d.dz; sa = H2; ca = c; Bp = action; sb = 1; cbb = 0; break;      // dwa = sum dz * a
            case 9: Ap = d.dz; sa = H2; ca = c; break;                                    // dba
            default: break;
        }
        // the quantity's tensor (tt_mlp_weights order: w1 b1 g1 be1 w2 b2 g2 be2 w3 b3 wa ba), its gradient row and optimizer
        // state: ONE run-time index into the kernel-argument tables (a per-case choice of the pointers made the compiler keep
        // all 40 of them live in scalar registers: > 1000 spill instructions)
        const int tensor = cat < 0 ? 0 : (cat < 3 ? cat + 5 : (cat < 6 ? cat - 2 : cat + 2));
        float *const outp = reinterpret_cast<float *const *>(&G)[tensor];
        AdamPtrs ad{nullptr, nullptr, nullptr, nullptr};
        if (A.on) ad = AdamPtrs{A.p[tensor], A.m[tensor], A.v[tensor], A.tgt[tensor]};
        // rows in flight per wave: two rounds at 256 rows (these workgroups do little else); four with row factors, which keeps
        // the launch at <= 168 registers = three workgroups per CU
        constexpr int RCH = ROWSCALE ? 16 : 32;
        const int rows = (n + 3) / 4, lo = wave * rows, hi = min(n, lo + rows);
        // unconditional loads from clamped addresses (see the dW2 blocks); a quantity without a second operand reads its first
        // one twice (same lines) and multiplies by 1
        const bool has_b = Bp != nullptr;
        const int cc = valid ? c : 0;
        const float *pa = Ap + (sa == 1 ? 0 : cc), *pb = has_b ? Bp + (sb == 1 ? 0 : cc) : pa;
        float t[RCH], u2[RCH];
        // row strides in VECTOR registers: as scalars the compiler forms all 2 x 32 products u * stride up front and keeps
        // them live in (then spilled) scalar registers
        int sav = sa, sbv = has_b ? sb : sa;
        asm volatile("" : "+v"(sav), "+v"(sbv));
        auto load_rows = [&](const int b0) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < RCH; ++u) {
                const int bc = min(b0 + u, n - 1);
                t[u] = pa[bc * sav];
                u2[u] = pb[bc * sbv];
            }
#pragma unroll
            for (int u = 0; u < RCH; ++u) {
                if (!(valid && b0 + u < hi)) t[u] = 0.f;
                if (!has_b) u2[u] = 1.f;
            }
        };
        load_rows(lo);
        AdamElem e0{};
        if (A.on && wave == 0 && valid) e0 = adam_load(ad, (size_t)c);
        fill_factors();
        float acc = 0.f;
        for (int b0 = lo; b0 < hi; b0 += RCH) {
            if (b0 != lo) load_rows(b0);
            // products rounded, then added in row order (what (dz * xh2).sum(0) does in the reference; written with the
            // _rn forms so that the compiler contracts none of them into an fma: the sums then do not depend on its choices)
#pragma unroll
            for (int u = 0; u < RCH; ++u)
                acc = __fadd_rn(acc, __fmul_rn(ROWSCALE ? __fmul_rn(t[u], row_factor(min(b0 + u, n - 1))) : t[u], u2[u]));
        }
        part[wave][0][lane] = acc;
        float bc1, sqrt_bc2;
        bias_corrections(bc1, sqrt_bc2);
        lds_barrier();
        if (wave == 0 && valid) {
            const float total = ((part[0][0][lane] + part[1][0][lane]) + part[2][0][lane]) + part[3][0][lane];
            st_out(&outp[c], total);
            if (A.on) adam_finish(A, ad, (size_t)c, total, e0, bc1, sqrt_bc2);
        }
        STAMPB(6, NU2 + NU1);
#ifdef TT_STAMPS
        __syncthreads(); if (threadIdx.x == 0) g_blk[blockIdx.x][1] = wall_clock64();
#endif
    }
    KEND(ROWSCALE ? 4 : 2);
}

// (three workgroups per CU: beside the policy's grid ~85 CUs are free for the ~205 of this launch -- 168 registers, see above)
template <bool ROWSCALE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_bwd_weights(const int n, const int critic, const float *__restrict__ obs,
                                                     const float *__restrict__ action, const Saved sv,
                                                     const BwdOut d, const Grads G, const AdamFused A, const RowScale RS) {
    __shared__ __attribute__((aligned(16))) float part[4][4][256];     // [wave][tile][lane*4 + r]
    __shared__ float f_s[ROWSCALE ? MAXB : 1];                         // the rows' factors, computed once per workgroup
    __shared__ __attribute__((aligned(16))) _Float16 stage_s[4 * 1024];   // forward-image pieces of a dW2 workgroup's patch
    kernarg_warm<24 + (int)sizeof(Saved) + (int)sizeof(BwdOut) + (int)sizeof(Grads) + (int)sizeof(AdamFused) + (int)sizeof(RowScale)>();
    bwd_weights_body<ROWSCALE, false>(blockIdx.x, n, critic, obs, action, sv, d, G, A, RS, part, f_s, TailSync{nullptr, nullptr, 0, nullptr}, 0,
                                      stage_s);
}

// learn()'s last two launches in ONE grid (the single-rank chain where the policy launch is small or learn() repeats per step):
// workgroups [0, nb) are k_fwd_small<critic> on (s, mu(s)) -- Q(s, mu(s)) and dQ/da through the UPDATED critic (DDPG_agent.py:100-103)
// --, the rest are k_bwd_weights<actor>, which request their operands, optimizer state and step count at once as always and then
// wait for the row workgroups' dQ/da in device memory (TailSync) instead of behind a launch boundary: the boundary (1.5-1.9 us),
// the kernel entry and the operand round trip of the weight-gradient launch leave learn()'s chain.  The row workgroups are
// dispatched first, so they never wait for a CU behind the workgroups that wait for them.  512 threads per workgroup (the row
// kernel's geometry); a weight-gradient workgroup uses the first 256.  LDS: the row kernel's tiles and the weight kernel's share it.
__global__ __launch_bounds__(64 * NW) void k_actor_tail(const int n, const float *__restrict__ obs, const float *__restrict__ mu,
                                                        const Weights Wc, float *__restrict__ q_out, float *__restrict__ dq_da,
                                                        const Saved sv, const BwdOut d, const Grads G, const AdamFused A,
                                                        const RowScale RS, const TailSync ts) {
    __shared__ __attribute__((aligned(16))) float lds[H1S_FLOATS + TR * DS + H1 * IN];
    const int nb = (n + TR - 1) / TR;
    if ((int)blockIdx.x < nb) {
        const Saved none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        const long long epoch = *A.step_dev;
        KBEGIN(3);
        fwd_small_body<true>(n, obs, mu, Wc, q_out, none, dq_da, nullptr, lds, lds + H1S_FLOATS, lds + H1S_FLOATS + TR * DS,
                             blockIdx.x * TR, nullptr, false, 0.f, 0.f, ts.rows, (unsigned)epoch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its rows' words have been sent
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ts.hints + blockIdx.x, (int)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        KEND(3);
        return;
    }
    if (threadIdx.x >= 256) return;
    static_assert(sizeof(float) * (H1S_FLOATS + TR * DS + H1 * IN) >= sizeof(float) * (4 * 4 * 256 + MAXB + 2048), "the weight kernel's LDS fits");
    float (&part)[4][4][256] = *reinterpret_cast<float (*)[4][4][256]>(lds);
    const long long epoch = *A.step_dev;
    bwd_weights_body<true, true>((int)blockIdx.x - nb, n, 0, obs, nullptr, sv, d, G, A, RS, part, lds + 4 * 4 * 256, ts, epoch,
                                 reinterpret_cast<_Float16 *>(lds + 4 * 4 * 256 + MAXB));
}

// ------------------------------------------------------------------------------------------------------
// torch.optim.Adam (amsgrad off, weight decay added to the gradient: networks.py:49-50,133) for every parameter
// tensor of a net in one launch, followed by the soft target update theta' <- theta' + tau*(theta - theta')
// (DDPG_agent.py:108-131).  step_dev holds the 1-based step count of THIS update.
constexpr int MAXT = 12;
struct AdamTable {
    float *p[MAXT], *m[MAXT], *v[MAXT], *tgt[MAXT];
    const float *g[MAXT];
    int numel[MAXT], block_start[MAXT + 1];
    int count;
    _Float16 *img_p, *img_t;      // fc2 images kept current for tensor 4 (w2 [300,400]) and its target (or nullptr)
};

__global__ __launch_bounds__(256) void k_adam_soft(const AdamTable T, const long long *__restrict__ step_dev,
                                                   const float lr, const float beta1, const float beta2,
                                                   const float eps, const float weight_decay, const float tau,
                                                   const float *__restrict__ bias_corr) {
    int ti = 0;
    while (ti + 1 < T.count && (int)blockIdx.x >= T.block_start[ti + 1]) ++ti;
    const int i = ((int)blockIdx.x - T.block_start[ti]) * 256 + threadIdx.x;
    if (i >= T.numel[ti]) return;
    float bcc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias_corr) {
#pragma unroll
        for (int q = 0; q < 5; ++q) bcc[q] = bias_corr[q];
    }
    float bc1, bc2;
    adam_bias_corrections(beta1, beta2, *step_dev, bias_corr != nullptr, bcc, bc1, bc2);
    float p = T.p[ti][i];
    const float g = fmaf(weight_decay, p, T.g[ti][i]);
    const float m = fmaf(beta1, T.m[ti][i], (1.f - beta1) * g);          // exp_avg.lerp_(grad, 1 - beta1)
    const float v = fmaf(beta2, T.v[ti][i], (1.f - beta2) * g * g);
    T.m[ti][i] = m;
    T.v[ti][i] = v;
    const float denom = sqrtf(v) / sqrtf(bc2) + eps;
    p -= (lr / bc1) * (m / denom);
    T.p[ti][i] = p;
    float tg = 0.f;
    if (T.tgt[ti]) {
        tg = T.tgt[ti][i];
        tg = fmaf(tau, p - tg, tg);
        T.tgt[ti][i] = tg;
    }
    if (ti == 4 && (T.img_p || T.img_t)) {
        const int nn = i / H1, k = i - nn * H1;
        if (T.img_p) img_store(T.img_p, nn, k, p, true);
        if (T.img_t && T.tgt[ti]) img_store(T.img_t, nn, k, tg, false);
    }
}

// k_adam_soft for data-parallel ranks WITHOUT a collective launch in front of it (include/ttenv.h: tt_p2p_*): the gradient of
// element i is the mean over the ranks of G_r[i], read from every rank's exchange block (this rank's own included) and summed
// in rank order -- the same bits on every rank.  A workgroup takes EPT x 256 consecutive elements of one tensor (few, fat
// workgroups: each of them waits and acquires once).  Hand-over (per site, epoch = this learn step's number):
//   publish  the launch's FIRST workgroup: system-scope release, then epoch -> word [site][me] of every rank's block.  The launch
//            that wrote G_me is the one in front of this one on its stream, so it is complete and written back by now;
//   wait     every workgroup: until the words [site][0..world) of its OWN block hold the epoch (relaxed system-scope loads of local
//            fine-grained memory, bounded), then a system-scope acquire;
//   read     G_r[i] by system-scope loads (never from a cache that could hold the previous step's value of the same address).
#ifndef TT_P2P_EPT
#define TT_P2P_EPT 4
#endif
constexpr int P2P_EPT = TT_P2P_EPT;
__global__ __launch_bounds__(256) void k_adam_soft_p2p(const AdamTable T, const ttp2p::Args X, const long long *__restrict__ step_dev,
                                                       const float lr, const float beta1, const float beta2, const float eps,
                                                       const float weight_decay, const float tau,
                                                       const float *__restrict__ bias_corr) {
    const long long step = *step_dev;
    int ti = 0;
    while (ti + 1 < T.count && (int)blockIdx.x >= T.block_start[ti + 1]) ++ti;
    const int base = ((int)blockIdx.x - T.block_start[ti]) * (256 * P2P_EPT) + threadIdx.x;
    const int n = T.numel[ti];
    // everything that does not depend on the exchange is requested first -- this rank's parameters, Adam moments, targets, bias
    // corrections -- so that those loads fly while the workgroup waits for the arrival words
    float pv[P2P_EPT], mv[P2P_EPT], vv[P2P_EPT], tv[P2P_EPT];
    const bool has_t = T.tgt[ti] != nullptr;
#pragma unroll
    for (int e = 0; e < P2P_EPT; ++e) {
        const int i = min(base + e * 256, n - 1);
        pv[e] = T.p[ti][i]; mv[e] = T.m[ti][i]; vv[e] = T.v[ti][i];
        tv[e] = has_t ? T.tgt[ti][i] : 0.f;
    }
    float bcc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias_corr) {
#pragma unroll
        for (int q = 0; q < 5; ++q) bcc[q] = bias_corr[q];
    }
    if (threadIdx.x == 0) {
        const int epoch = (int)step;
        if (blockIdx.x == 0) {
#ifndef TT_P2P_NOFENCE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
            for (int r = 0; r < X.world; ++r)
                __hip_atomic_store(X.arrive[r] + X.site * ttp2p::MAXR + X.me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const int *mine = X.arrive[X.me] + X.site * ttp2p::MAXR;
        const unsigned long long t0 = wall_clock64();
        bool gave_up = false;
        for (int r = 0; r < X.world && !gave_up; ++r) {
            while (__hip_atomic_load(mine + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch < 0) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > X.wait_ticks) {      // never hang: mark (host-visible) and go on; the caller treats the ranks as diverged
                    __hip_atomic_store(X.gave_up_host, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    gave_up = true;
                    break;
                }
            }
        }
#ifndef TT_P2P_NOFENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#endif
    }
    __syncthreads();
    const size_t goff = X.tensor_offset[ti];           // the tensor's place (floats) in the site's flat buffer
    float gsum[P2P_EPT];
#pragma unroll
    for (int e = 0; e < P2P_EPT; ++e) gsum[e] = 0.f;
    for (int r = 0; r < X.world; ++r) {                 // rank order: the same sum on every rank
        const float *src = X.grad[r] + goff;
        float part[P2P_EPT];
#pragma unroll
        for (int e = 0; e < P2P_EPT; ++e) {
            const int i = base + e * 256;
            // (this rank's own buffer: an ordinary load -- its backward launch is over; a peer's: system scope, served by the
            // owner's memory, never by a cache of this GPU that could still hold last step's value of the same address)
            part[e] = i >= n ? 0.f : (r == X.me ? src[i] : __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
#pragma unroll
        for (int e = 0; e < P2P_EPT; ++e) gsum[e] = __fadd_rn(gsum[e], part[e]);
    }
    float bc1, bc2;
    adam_bias_corrections(beta1, beta2, step, bias_corr != nullptr, bcc, bc1, bc2);
    const float sqrt_bc2 = sqrtf(bc2);
    const float world_f = (float)X.world;
#pragma unroll
    for (int e = 0; e < P2P_EPT; ++e) {
        const int i = base + e * 256;
        if (i >= n) continue;
        const float grad = X.world > 1 ? __fdiv_rn(gsum[e], world_f) : gsum[e];
        float p = pv[e];
        const float g = fmaf(weight_decay, p, grad);
        const float m = fmaf(beta1, mv[e], (1.f - beta1) * g);          // exp_avg.lerp_(grad, 1 - beta1)
        const float v = fmaf(beta2, vv[e], (1.f - beta2) * g * g);
        T.m[ti][i] = m;
        T.v[ti][i] = v;
        const float denom = sqrtf(v) / sqrt_bc2 + eps;
        p -= (lr / bc1) * (m / denom);
        T.p[ti][i] = p;
        float tg = 0.f;
        if (has_t) {
            tg = tv[e];
            tg = fmaf(tau, p - tg, tg);
            T.tgt[ti][i] = tg;
        }
        if (ti == 4 && (T.img_p || T.img_t)) {
            const int nn = i / H1, k = i - nn * H1;
            if (T.img_p) img_store(T.img_p, nn, k, p, true);
            if (T.img_t && has_t) img_store(T.img_t, nn, k, tg, false);
        }
    }
}

// the image of fc2 from scratch (tt_mlp_fc2_image_pack): one thread per weight; the padding of the buffer is never written
__global__ __launch_bounds__(256) void k_img_pack(const float *__restrict__ w2, _Float16 *__restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H2 * H1) return;
    const int nn = i / H1, k = i - nn * H1;
    img_store(img, nn, k, w2[i], true);
}

// y = r + gamma * q' * (1 - done) (DDPG_agent.py:89-93) and the learn-step counter
__global__ void k_td_target(const int n, const float *__restrict__ r, const float *__restrict__ q_next,
                            const uint8_t *__restrict__ done, const float gamma, float *__restrict__ y,
                            long long *__restrict__ step_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && step_dev) *step_dev += 1;
    if (i < n) y[i] = done[i] ? r[i] : fmaf(gamma, q_next[i], r[i]);
}

// the rest of the critic once the action is known (networks.py:62-68) and the TD target (DDPG_agent.py:89-93) in one
// launch: q'[b] = q(relu(z_state[b] + action_value(a[b]))), y[b] = r[b] + gamma * q'[b] * (1 - done[b]).  16 lanes per row.
__global__ __launch_bounds__(256) void k_head_td(const int n, const float *__restrict__ z_state,
                                                 const float *__restrict__ action, const Weights W,
                                                 const float *__restrict__ r, const uint8_t *__restrict__ done,
                                                 const float gamma, float *__restrict__ y, float *__restrict__ q_out,
                                                 long long *__restrict__ step_dev) {
    const int tid = threadIdx.x, l15 = tid & 15, row = blockIdx.x * 16 + (tid >> 4);
    if (blockIdx.x == 0 && tid == 0 && step_dev) *step_dev += 1;
    const bool ok = row < n;
    const float a = ok ? action[row] : 0.f;
    float dot = 0.f;
    if (ok) {
        float z[19], wa[19], ba[19], w3[19];
#pragma unroll
        for (int i = 0; i < 19; ++i) {
            const int col = l15 + 16 * i;
            const bool real = col < H2;
            z[i] = real ? z_state[(size_t)row * H2 + col] : 0.f;
            wa[i] = real ? W.wa[col] : 0.f; ba[i] = real ? W.ba[col] : 0.f; w3[i] = real ? W.w3[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 19; ++i) dot = fmaf(fmaxf(z[i] + fmaf(a, wa[i], ba[i]), 0.f), w3[i], dot);
    }
    dot = row_sum16(dot);
    if (ok && l15 == 0) {
        const float q = dot + W.b3[0];
        if (q_out) q_out[row] = q;
        y[row] = done[row] ? r[row] : fmaf(gamma, q, r[row]);
    }
}

Weights to_weights(const tt_mlp_weights *w) {
    return Weights{w->w1, w->b1, w->g1, w->be1, w->w2, w->b2, w->g2, w->be2, w->w3, w->b3, w->wa, w->ba,
                   reinterpret_cast<const _Float16 *>(w->fc2_img)};
}

bool ok_shape(const tt_mlp_weights *w, bool critic) {
    return w && w->in_dim == IN && w->fc1_dims == H1 && w->fc2_dims == H2 && w->w1 && w->b1 && w->g1 && w->be1 && w->w2 &&
           w->b2 && w->g2 && w->be2 && w->w3 && w->b3 && (!critic || (w->wa && w->ba));
}

}  // namespace

extern "C" {

int tt_mlp_forward_save(int n, int critic, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                        const tt_mlp_saved *saved, float *dq_da, tt_stream_t stream) {
    if (n < 0 || !obs || !out || !ok_shape(w, critic != 0) || (critic && !action)) return TT_EINVAL;
    if (n == 0) return TT_OK;
    Saved sv{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (saved) {
        if (!saved->xh1 || !saved->h1 || !saved->xh2 || !saved->h2 || !saved->rstd1 || !saved->rstd2) return TT_EINVAL;
        sv = Saved{saved->xh1, saved->h1, saved->xh2, saved->h2, saved->rstd1, saved->rstd2};
    }
    const dim3 grid((n + TR - 1) / TR), block(64 * NW);
    if (critic)
        hipLaunchKernelGGL(k_fwd_small<true>, grid, block, 0, stream, n, obs, action, w->w1, w->b1, w->g1, w->be1, to_weights(w), out, sv, dq_da,
                           static_cast<float *>(nullptr));
    else
        hipLaunchKernelGGL(k_fwd_small<false>, grid, block, 0, stream, n, obs, action, w->w1, w->b1, w->g1, w->be1, to_weights(w), out, sv,
                           static_cast<float *>(nullptr), static_cast<float *>(nullptr));
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

static int forward_multi_impl(int n, int count, const tt_fwd_job *jobs, const tt_sample_args *sample, int64_t *k_snapshot,
                              tt_stream_t stream) {
    if (n < 0 || count < 1 || count > 4 || !jobs) return TT_EINVAL;
    if (n == 0) return TT_OK;
    FwdJobs J{};
    J.n = n;
    J.blocks_per_job = (n + TR - 1) / TR;
    J.write_s = J.write_s2 = -1;
    if (sample) {
        if (sample->batch != n) return TT_EINVAL;
        const int rc = ttnet::make_ring_sample(sample, J.R);
        if (rc != TT_OK) return rc;
        J.sampled = 1;
        J.R.progress = const_cast<int *>(sample->step_progress);
        J.k_snapshot = reinterpret_cast<long long *>(k_snapshot);
    }
    for (int i = 0; i < count; ++i) {
        const tt_fwd_job &q = jobs[i];
        const bool critic = q.critic != 0;
        if (!q.obs || !ok_shape(q.w, critic) || (critic && !q.action && !q.z_state) || (!q.out && !q.z_state)) return TT_EINVAL;
        if (sample) {      // every job reads the draw's s or s'; a critic job's action is the draw's a
            if (q.obs != sample->s_out && q.obs != sample->s2_out) return TT_EINVAL;
            if (q.action && (q.action != sample->a_out || q.obs != sample->s_out)) return TT_EINVAL;
            if (q.obs == sample->s_out && J.write_s < 0) J.write_s = i;
            if (q.obs == sample->s2_out && J.write_s2 < 0) J.write_s2 = i;
        }
        Saved sv{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (q.saved) {
            const tt_mlp_saved *p = q.saved;
            if (!p->xh1 || !p->h1 || !p->xh2 || !p->h2 || !p->rstd1 || !p->rstd2) return TT_EINVAL;
            sv = Saved{p->xh1, p->h1, p->xh2, p->h2, p->rstd1, p->rstd2};
        }
        J.j[i] = FwdJob{q.obs, q.action, to_weights(q.w), q.out, sv, q.dq_da, critic ? q.z_state : nullptr, critic ? 1 : 0};
    }
    if (sample && (J.write_s < 0 || J.write_s2 < 0)) return TT_EINVAL;      // the later launches need all five batch buffers
    hipLaunchKernelGGL(k_fwd_multi, dim3(count * J.blocks_per_job), dim3(64 * NW), 0, stream, J);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_mlp_forward_multi(int n, int count, const tt_fwd_job *jobs, tt_stream_t stream) {
    return forward_multi_impl(n, count, jobs, nullptr, nullptr, stream);
}

int tt_mlp_forward_multi_sampled(int n, int count, const tt_fwd_job *jobs, const tt_sample_args *sample, int64_t *k_snapshot,
                                 tt_stream_t stream) {
    if (!sample) return TT_EINVAL;
    return forward_multi_impl(n, count, jobs, sample, k_snapshot, stream);
}

int tt_critic_state_forward(int n, const float *obs, const tt_mlp_weights *w, float *z_state, tt_stream_t stream) {
    if (n < 0 || !obs || !z_state || !ok_shape(w, true)) return TT_EINVAL;
    if (n == 0) return TT_OK;
    const Saved sv{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_fwd_small<true>, dim3((n + TR - 1) / TR), dim3(64 * NW), 0, stream, n, obs,
                       static_cast<const float *>(nullptr), w->w1, w->b1, w->g1, w->be1, to_weights(w), static_cast<float *>(nullptr), sv,
                       static_cast<float *>(nullptr), z_state);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_critic_head_td(int n, const float *z_state, const float *action, const tt_mlp_weights *w, const float *reward,
                      const uint8_t *done, float gamma, float *y, float *q_out, int64_t *step_dev, tt_stream_t stream) {
    if (n <= 0 || !z_state || !action || !ok_shape(w, true) || !reward || !done || !y) return TT_EINVAL;
    hipLaunchKernelGGL(k_head_td, dim3((n + 15) / 16), dim3(256), 0, stream, n, z_state, action, to_weights(w), reward, done,
                       gamma, y, q_out, reinterpret_cast<long long *>(step_dev));
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

static int backward_impl(int n, int critic, int mode, float scale, const float *obs, const float *action, const float *d_out,
                         const float *out, const float *y, const float *aux, const tt_mlp_weights *w,
                         const tt_mlp_saved *saved, const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, const AdamFused &A,
                         const tt_td_input *tdi, tt_stream_t stream) {
    if (n <= 0 || !obs || !out || !ok_shape(w, critic != 0) || !ok_shape(grads, critic != 0) || !saved || !ws ||
        (critic && !action) || mode < 0 || mode > 2 || (mode == 0 && !d_out) || (mode == 1 && !y && !tdi) || (mode == 2 && !aux))
        return TT_EINVAL;
    TdIn td{};
    if (tdi) {
        const tt_mlp_weights *tw = tdi->target_critic;
        if (!critic || mode != 1 || !tdi->z_state || !tdi->mu_target || !ok_shape(tw, true) || !tdi->reward || !tdi->done ||
            !tdi->y_out)
            return TT_EINVAL;
        td = TdIn{tdi->z_state, tdi->mu_target, tdi->reward, tdi->done, tw->wa, tw->ba, tw->w3, tw->b3, tdi->gamma,
                  tdi->y_out, tdi->q_out, reinterpret_cast<long long *>(tdi->step_dev),
                  reinterpret_cast<long long *>(tdi->window_dev), nullptr, 0.f, 0.f, 0};
    }
    if (!saved->xh1 || !saved->h1 || !saved->xh2 || !saved->h2 || !saved->rstd1 || !saved->rstd2 || !ws->dpre || !ws->dz ||
        !ws->dx2 || !ws->dy1 || !ws->dx1)
        return TT_EINVAL;
    const Saved sv{saved->xh1, saved->h1, saved->xh2, saved->h2, saved->rstd1, saved->rstd2};
    const BwdOut o{ws->dpre, ws->dz, ws->dx2, ws->dy1, ws->dx1};
    const dim3 grid((n + TR - 1) / TR), block(256), block_rows(64 * NW);
    if (critic)
        hipLaunchKernelGGL(k_bwd_rows<true>, grid, block_rows, 0, stream, n, mode, scale, d_out, out, y, aux,
                           to_weights(w), sv, o, td);
    else
        hipLaunchKernelGGL(k_bwd_rows<false>, grid, block_rows, 0, stream, n, mode, scale, d_out, out, y, aux,
                           to_weights(w), sv, o, td);
    if (hipGetLastError() != hipSuccess) return TT_EHIP;
    const Grads G{const_cast<float *>(grads->w1), const_cast<float *>(grads->b1), const_cast<float *>(grads->g1),
                  const_cast<float *>(grads->be1), const_cast<float *>(grads->w2), const_cast<float *>(grads->b2),
                  const_cast<float *>(grads->g2), const_cast<float *>(grads->be2), const_cast<float *>(grads->w3),
                  const_cast<float *>(grads->b3), const_cast<float *>(grads->wa), const_cast<float *>(grads->ba)};
    const int sum_blocks = critic ? SUMB_CRITIC : SUMB_ACTOR;
    const RowScale none{nullptr, nullptr, 1.f};
    hipLaunchKernelGGL(k_bwd_weights<false>, dim3(NU2 + NU1 + sum_blocks), block, 0, stream, n, critic, obs, action, sv, o, G, A, none);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

static bool saved_ok(const tt_mlp_saved *s) { return s && s->xh1 && s->h1 && s->xh2 && s->h2 && s->rstd1 && s->rstd2; }
static bool ws_ok(const tt_mlp_bwd_ws *w) { return w && w->dpre && w->dz && w->dx2 && w->dy1 && w->dx1; }
static Grads to_grads(const tt_mlp_weights *g) {
    return Grads{const_cast<float *>(g->w1), const_cast<float *>(g->b1), const_cast<float *>(g->g1), const_cast<float *>(g->be1),
                 const_cast<float *>(g->w2), const_cast<float *>(g->b2), const_cast<float *>(g->g2), const_cast<float *>(g->be2),
                 const_cast<float *>(g->w3), const_cast<float *>(g->b3), const_cast<float *>(g->wa), const_cast<float *>(g->ba)};
}

int tt_mlp_backward(int n, int critic, int mode, float scale, const float *obs, const float *action, const float *d_out,
                    const float *out, const float *y, const float *aux, const tt_mlp_weights *w,
                    const tt_mlp_saved *saved, const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, const tt_td_input *td,
                    tt_stream_t stream) {
    AdamFused A{};
    return backward_impl(n, critic, mode, scale, obs, action, d_out, out, y, aux, w, saved, ws, grads, A, td, stream);
}

int tt_mlp_backward_adam(int n, int critic, int mode, float scale, const float *obs, const float *action, const float *d_out,
                         const float *out, const float *y, const float *aux, const tt_mlp_weights *w,
                         const tt_mlp_saved *saved, const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, int count,
                         float *const *params, float *const *exp_avg, float *const *exp_avg_sq, float *const *targets,
                         const int64_t *step_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                         float tau, const tt_td_input *td, tt_stream_t stream) {
    if (count != (critic ? 12 : 10) || !params || !exp_avg || !exp_avg_sq || !step_dev) return TT_EINVAL;
    AdamFused A{};
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !exp_avg[i] || !exp_avg_sq[i]) return TT_EINVAL;
        A.p[i] = params[i]; A.m[i] = exp_avg[i]; A.v[i] = exp_avg_sq[i]; A.tgt[i] = targets ? targets[i] : nullptr;
    }
    A.step_dev = reinterpret_cast<const long long *>(step_dev);
    A.lr = lr; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.weight_decay = weight_decay; A.tau = tau;
    A.on = 1;
    return backward_impl(n, critic, mode, scale, obs, action, d_out, out, y, aux, w, saved, ws, grads, A, td, stream);
}

int tt_mlp_backward_rows_pair(int n, float scale_critic, const float *q_out, const tt_mlp_weights *critic,
                              const tt_mlp_saved *saved_critic, const tt_mlp_bwd_ws *ws_critic, const tt_td_input *tdi,
                              const float *mu_out, const tt_mlp_weights *actor, const tt_mlp_saved *saved_actor,
                              const tt_mlp_bwd_ws *ws_actor, const tt_image_job *image, tt_stream_t stream) {
    ImageJob ij{};
    if (image) {
        const tt_mlp_weights *w = image->actor;
        const tt_ring_cursor *c = image->cursor;
        if (!ok_shape(w, false) || !w->split_ws || !c || !c->cursor || !c->k_dev || c->slots <= 0 || (tdi && (c->k_dev == tdi->step_dev ||
                                                                                                         c->k_dev == tdi->window_dev)))
            return TT_EINVAL;      // (the cursor's step number must be a word this launch does not advance)
        ij.W = ttnet::to_weights(w);
        ij.ws = reinterpret_cast<unsigned char *>(w->split_ws);
        ij.ws_alt = reinterpret_cast<unsigned char *>(w->split_ws_alt);
        ij.cur = ttnet::RingCursor{reinterpret_cast<const long long *>(c->k_dev), c->slots, c->cursor};
        ij.on = 1;
    }
    if (n <= 0 || !q_out || !mu_out || !ok_shape(critic, true) || !ok_shape(actor, false) || !saved_ok(saved_critic) ||
        !saved_ok(saved_actor) || !ws_ok(ws_critic) || !ws_ok(ws_actor) || !tdi || ws_critic->dx2 == ws_actor->dx2)
        return TT_EINVAL;
    const tt_mlp_weights *tw = tdi->target_critic;
    if (!tdi->z_state || !tdi->mu_target || !ok_shape(tw, true) || !tdi->reward || !tdi->done || !tdi->y_out) return TT_EINVAL;
    const TdIn td{tdi->z_state, tdi->mu_target, tdi->reward, tdi->done, tw->wa, tw->ba, tw->w3, tw->b3, tdi->gamma,
                  tdi->y_out, tdi->q_out, reinterpret_cast<long long *>(tdi->step_dev),
                  reinterpret_cast<long long *>(tdi->window_dev), tdi->bias_corr_out, tdi->adam_beta1, tdi->adam_beta2, 1};
    const Saved sc{saved_critic->xh1, saved_critic->h1, saved_critic->xh2, saved_critic->h2, saved_critic->rstd1, saved_critic->rstd2};
    const Saved sa{saved_actor->xh1, saved_actor->h1, saved_actor->xh2, saved_actor->h2, saved_actor->rstd1, saved_actor->rstd2};
    const BwdOut oc{ws_critic->dpre, ws_critic->dz, ws_critic->dx2, ws_critic->dy1, ws_critic->dx1};
    const BwdOut oa{ws_actor->dpre, ws_actor->dz, ws_actor->dx2, ws_actor->dy1, ws_actor->dx1};
    hipLaunchKernelGGL(k_bwd_rows_pair, dim3(2 * ((n + TR - 1) / TR) + 1 + (ij.on ? IMAGE_WGS : 0)), dim3(64 * NW), 0, stream, n,
                       scale_critic, q_out, to_weights(critic), sc, oc, td, mu_out, to_weights(actor), sa, oa, ij);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_mlp_backward_weights(int n, int critic, const float *obs, const float *action, const tt_mlp_saved *saved,
                            const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, const float *row_dq_da, const float *row_mu,
                            float row_scale, int count, float *const *params, float *const *exp_avg, float *const *exp_avg_sq,
                            float *const *targets, const int64_t *step_dev, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float tau, const tt_fc2_images *images, const float *bias_corr,
                            tt_stream_t stream) {
    if (n <= 0 || !obs || (critic && !action) || !saved_ok(saved) || !ws_ok(ws) || !ok_shape(grads, critic != 0) ||
        ((row_dq_da == nullptr) != (row_mu == nullptr)))
        return TT_EINVAL;
    AdamFused A{};
    if (count) {        // Adam + soft update applied in the same launch (one rank); count = 0: gradients only
        if (count != (critic ? 12 : 10) || !params || !exp_avg || !exp_avg_sq || !step_dev) return TT_EINVAL;
        for (int i = 0; i < count; ++i) {
            if (!params[i] || !exp_avg[i] || !exp_avg_sq[i]) return TT_EINVAL;
            A.p[i] = params[i]; A.m[i] = exp_avg[i]; A.v[i] = exp_avg_sq[i]; A.tgt[i] = targets ? targets[i] : nullptr;
        }
        A.step_dev = reinterpret_cast<const long long *>(step_dev);
        A.lr = lr; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.weight_decay = weight_decay; A.tau = tau;
        A.on = 1;
        A.bias_corr = bias_corr;
        if (images) {
            A.img_p = reinterpret_cast<_Float16 *>(images->net);
            A.img_t = reinterpret_cast<_Float16 *>(images->target);
        }
    }
    const Saved sv{saved->xh1, saved->h1, saved->xh2, saved->h2, saved->rstd1, saved->rstd2};
    const BwdOut o{ws->dpre, ws->dz, ws->dx2, ws->dy1, ws->dx1};
    const RowScale rs{row_dq_da, row_mu, row_scale};
    const dim3 grid(NU2 + NU1 + (critic ? SUMB_CRITIC : SUMB_ACTOR));
    if (row_dq_da) {
        if (n > MAXB) return TT_EINVAL;
        hipLaunchKernelGGL(k_bwd_weights<true>, grid, dim3(256), 0, stream, n, critic, obs, action, sv, o, to_grads(grads), A, rs);
    } else {
        hipLaunchKernelGGL(k_bwd_weights<false>, grid, dim3(256), 0, stream, n, critic, obs, action, sv, o, to_grads(grads), A, rs);
    }
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_mlp_actor_tail(int n, const float *obs, const float *mu, const tt_mlp_weights *critic, float *q_out, float *dq_da,
                      const tt_mlp_saved *saved, const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, float row_scale, int count,
                      float *const *params, float *const *exp_avg, float *const *exp_avg_sq, float *const *targets,
                      const int64_t *step_dev, float lr, float beta1, float beta2, float eps, float weight_decay, float tau,
                      const tt_fc2_images *images, const float *bias_corr, int32_t *tail_words, int32_t *gave_up_host,
                      tt_stream_t stream) {
    if (n <= 0 || n > MAXB || !obs || !mu || !q_out || !dq_da || !ok_shape(critic, true) || !saved_ok(saved) || !ws_ok(ws) ||
        !ok_shape(grads, false) || count != 10 || !params || !exp_avg || !exp_avg_sq || !step_dev || !tail_words)
        return TT_EINVAL;
    const int nb = (n + TR - 1) / TR;
    if (nb > 64) return TT_EINVAL;                         // (one wave polls the producers' words)
    AdamFused A{};
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !exp_avg[i] || !exp_avg_sq[i]) return TT_EINVAL;
        A.p[i] = params[i]; A.m[i] = exp_avg[i]; A.v[i] = exp_avg_sq[i]; A.tgt[i] = targets ? targets[i] : nullptr;
    }
    A.step_dev = reinterpret_cast<const long long *>(step_dev);
    A.lr = lr; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.weight_decay = weight_decay; A.tau = tau;
    A.on = 1;
    A.bias_corr = bias_corr;
    if (images) {
        A.img_p = reinterpret_cast<_Float16 *>(images->net);
        A.img_t = reinterpret_cast<_Float16 *>(images->target);
    }
    const Saved sv{saved->xh1, saved->h1, saved->xh2, saved->h2, saved->rstd1, saved->rstd2};
    const BwdOut o{ws->dpre, ws->dz, ws->dx2, ws->dy1, ws->dx1};
    const RowScale rs{dq_da, mu, row_scale};
    // tail_words: [0, 64) the row workgroups' hints, [64, 64 + 2 n) the rows' 8-byte words
    const TailSync ts{tail_words, reinterpret_cast<unsigned long long *>(tail_words + 64), nb, gave_up_host};
    hipLaunchKernelGGL(k_actor_tail, dim3(nb + NU2 + NU1 + SUMB_ACTOR), dim3(64 * NW), 0, stream, n, obs, mu, to_weights(critic), q_out,
                       dq_da, sv, o, to_grads(grads), A, rs, ts);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_adam_soft_update(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                        float *const *exp_avg_sq, float *const *targets, const int32_t *numel, const int64_t *step_dev,
                        float lr, float beta1, float beta2, float eps, float weight_decay, float tau,
                        const tt_fc2_images *images, const float *bias_corr, tt_stream_t stream) {
    if (count <= 0 || count > MAXT || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !step_dev) return TT_EINVAL;
    AdamTable T{};
    if (images && (images->net || images->target)) {       // tensors must then be in tt_mlp_weights order: w2 is number 4
        if (count < 5 || numel[4] != H2 * H1) return TT_EINVAL;
        T.img_p = reinterpret_cast<_Float16 *>(images->net);
        T.img_t = reinterpret_cast<_Float16 *>(images->target);
    }
    T.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] <= 0) return TT_EINVAL;
        T.p[i] = params[i]; T.g[i] = grads[i]; T.m[i] = exp_avg[i]; T.v[i] = exp_avg_sq[i];
        T.tgt[i] = targets ? targets[i] : nullptr;
        T.numel[i] = numel[i];
        T.block_start[i] = blocks;
        blocks += (numel[i] + 255) / 256;
    }
    T.block_start[count] = blocks;
    hipLaunchKernelGGL(k_adam_soft, dim3(blocks), dim3(256), 0, stream, T, reinterpret_cast<const long long *>(step_dev), lr,
                       beta1, beta2, eps, weight_decay, tau, bias_corr);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_adam_soft_update_p2p(tt_p2p *x, int site, int count, float *const *params, float *const *exp_avg,
                            float *const *exp_avg_sq, float *const *targets, const int32_t *numel, const int64_t *step_dev,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float tau,
                            const tt_fc2_images *images, const float *bias_corr, tt_stream_t stream) {
    if (!x || site < 0 || site >= x->sites || count <= 0 || count > MAXT || !params || !exp_avg || !exp_avg_sq || !numel || !step_dev)
        return TT_EINVAL;
    AdamTable T{};
    ttp2p::Args X{};
    if (images && (images->net || images->target)) {       // tensors must then be in tt_mlp_weights order: w2 is number 4
        if (count < 5 || numel[4] != H2 * H1) return TT_EINVAL;
        T.img_p = reinterpret_cast<_Float16 *>(images->net);
        T.img_t = reinterpret_cast<_Float16 *>(images->target);
    }
    T.count = count;
    int blocks = 0;
    size_t off = 0;
    for (int i = 0; i < count; ++i) {
        if (!params[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] <= 0) return TT_EINVAL;
        T.p[i] = params[i]; T.m[i] = exp_avg[i]; T.v[i] = exp_avg_sq[i];
        X.tensor_offset[i] = (unsigned)off;
        T.tgt[i] = targets ? targets[i] : nullptr;
        T.numel[i] = numel[i];
        T.block_start[i] = blocks;
        blocks += (numel[i] + 256 * P2P_EPT - 1) / (256 * P2P_EPT);
        off += (size_t)numel[i];
    }
    T.block_start[count] = blocks;
    if (off != (size_t)x->numel[site]) return TT_EINVAL;      // the tensors must tile the site's buffer exactly
    X.world = x->world; X.me = x->rank; X.site = site;
    for (int r = 0; r < x->world; ++r) {
        if (!x->attached[r] || !x->block[r] || !x->flags[r]) return TT_EINVAL;      // every peer's blocks must have been opened (tt_p2p_attach)
        X.arrive[r] = reinterpret_cast<int *>(x->flags[r]);
        X.grad[r] = reinterpret_cast<const float *>(x->block[r] + x->offset[site]);
    }
    X.gave_up_host = x->gave_up_host;
    X.wait_ticks = x->wait_ticks;
    hipLaunchKernelGGL(k_adam_soft_p2p, dim3(blocks), dim3(256), 0, stream, T, X, reinterpret_cast<const long long *>(step_dev), lr,
                       beta1, beta2, eps, weight_decay, tau, bias_corr);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

uint64_t tt_mlp_fc2_image_bytes(void) { return (uint64_t)IMG_HALVES * 2; }

int tt_mlp_fc2_image_pack(const tt_mlp_weights *w, tt_stream_t stream) {
    if (!w || !w->w2 || !w->fc2_img || w->fc1_dims != H1 || w->fc2_dims != H2) return TT_EINVAL;
    hipLaunchKernelGGL(k_img_pack, dim3((H2 * H1 + 255) / 256), dim3(256), 0, stream, w->w2,
                       reinterpret_cast<_Float16 *>(w->fc2_img));
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

#ifdef TT_STAMPS
int tt_debug_blocks(unsigned long long *out1024) {
    return hipMemcpyFromSymbol(out1024, HIP_SYMBOL(g_blk), sizeof(unsigned long long) * 1024) == hipSuccess ? 0 : -3;
}
int tt_debug_wst(unsigned long long *out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_wst), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -3;
}
int tt_debug_kblocks(unsigned long long *out6x1024) {
    return hipMemcpyFromSymbol(out6x1024, HIP_SYMBOL(g_kblk), sizeof(unsigned long long) * 6 * 1024) == hipSuccess ? 0 : -3;
}
int tt_debug_log_learn(int k, unsigned long long *out, int reset) {
    if (k < 0 || k >= 5) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_log_learn), sizeof(TTLog), sizeof(TTLog) * (size_t)k) != hipSuccess) return -3;
    if (reset) { const unsigned long long z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_log_learn), &z, sizeof(z), sizeof(TTLog) * (size_t)k) != hipSuccess) return -3; }
    return 0;
}
int tt_debug_learn_poll(unsigned long long *out4) {
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(ttnet::g_poll), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -3;
}
int tt_debug_substamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sub), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
int tt_debug_stamps(unsigned long long *out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -3;
}
#endif

int tt_td_target(int n, const float *reward, const float *q_next, const uint8_t *done, float gamma, float *y,
                 int64_t *step_dev, tt_stream_t stream) {
    if (n <= 0 || !reward || !q_next || !done || !y) return TT_EINVAL;
    hipLaunchKernelGGL(k_td_target, dim3((n + 255) / 256), dim3(256), 0, stream, n, reward, q_next, done, gamma, y,
                       reinterpret_cast<long long *>(step_dev));
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

}  // extern "C"
