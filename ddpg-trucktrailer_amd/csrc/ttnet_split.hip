// ttnet_split.hip -- the N-env forward of the reference's actor / critic MLPs (DDPG/networks.py:55-68, 138-147) on the
// bf16 MFMA of gfx950 at f32 accuracy.
//
// Why: the f32-input MFMA runs at 1/16 of the bf16 rate (155 TFLOP/s vs ~2.5 PFLOP/s dense), and fc2 (400 -> 300) is
// 93 % of the network's arithmetic.  An f32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand
// bits, by truncation: x = hi + mid + lo with no rounding anywhere), and a product of two bf16 numbers is exact in
// f32, so
//     x*w = hi*hi' + (hi*mid' + mid*hi') + (mid*mid' + hi*lo' + lo*hi') + [mid*lo' + lo*mid' + lo*lo']
// where the bracket is below 2^-23 |x*w|: six bf16 MFMAs with f32 accumulation reproduce the f32 product to within
// its own rounding error, at 6/16 of the f32 MFMA's cycles.
//
// Orientation: everything is computed TRANSPOSED (neurons on the MFMA's M axis, envs on N), with
// v_mfma_f32_32x32x16_bf16.  Its 32x32 result has the env on the lane (lane & 31) and 16 neurons in the lane's
// registers, which is exactly the B-operand layout of the next product that sums over neurons: the layer-1 result
// goes through bias + LayerNorm + ReLU in registers and feeds layer 2 with no LDS round trip and no lane movement
// (the k order inside a step is permuted; the packed weights are laid out to match).  Both LayerNorms and the
// 300 -> 1 head reduce over registers plus ONE cross-lane exchange (lane ^ 32).
//
//   workgroup = 4 waves = 128 envs (32 per wave); per wave: h1 = 13 tiles x 16 registers (f32), layer-2
//   accumulators 10 tiles x 16 registers; fc2 is streamed once per workgroup, as pre-split bf16 planes in MFMA
//   fragment order, through a 3-deep LDS ring shared by the four waves (32 KB per k16 step), filled by LDS-DMA.
//   layer 1 (K = 23, 6 % of the arithmetic) runs on the exact f32 MFMA (32x32x2) from raw fc1 weights in LDS.
#include "ttnet_common.h"

namespace ttnet {
namespace {

#ifdef TT_STAMPS   // diagnostic build only: clocks of workgroup 0 / thread 0 at the phase boundaries
__device__ unsigned long long g_nstamps[16];
#define NSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_nstamps[i] = wall_clock64(); g_nstamps[8 + i] = clock64(); } } while (0)
#else
#define NSTAMP(i) do { } while (0)
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int ROWS = 128, WROWS = 32;       // envs per workgroup / per wave
constexpr int T1 = 13, H1P = 32 * T1;       // layer-1 neuron tiles (416 >= 400)
constexpr int T2 = 10, H2P = 32 * T2;       // layer-2 neuron tiles (320 >= 300)
#ifdef TT_DBG_STEPS                         // timing experiments only (wrong results): fewer k16 steps of layer 2
constexpr int STEPS = TT_DBG_STEPS;
#else
constexpr int STEPS = H1 / 16;              // 25 k16 steps of layer 2
#endif
constexpr int KQ = 12;                      // k2 steps of layer 1 (24 >= 23)
constexpr int PIECES = 32;                  // one k16 step of packed fc2 = 10 tiles x 3 planes = 30 pieces of 64 lanes x 16 B,
                                            // padded to 32 so that each of the 4 waves moves exactly 8 pieces
constexpr int CHUNK_U4 = PIECES * 64;
constexpr int CHUNK_BYTES = CHUNK_U4 * 16;  // 32,768
constexpr int RING = 3;
constexpr int RING_BYTES = RING * CHUNK_BYTES;                        // 98,304
constexpr int LDS_BYTES = RING_BYTES + (3 * H1P + 6 * H2P + H1 * IN) * 4;   // + 4,992 + 7,680 + 36,800 = 147,776

// x0, x1 -> their three bf16 pieces, packed as (x0 piece | x1 piece << 16).  Truncation, so every piece and every
// remainder is exact.
__device__ __forceinline__ void split2(const float x0, const float x1, uint32_t &ph, uint32_t &pm, uint32_t &pl) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    ph = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    pm = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    pl = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

__device__ __forceinline__ f32x16 mfma_bf16(const uint4 a, const uint4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// fc2 [300,400] f32 -> packed[s][u][plane][lane] (16 B each): lane (r = lane & 31, h = lane >> 5) of neuron tile u
// holds, for k16 step s, the weights of neuron 32u + r at inputs k(j) = 16 s + 8 (j >> 2) + 4 h + (j & 3), j = 0..7 --
// the order in which a 32x32 accumulator tile presents its rows when it is used as the other operand.
__global__ __launch_bounds__(256) void k_split_pack(const float *__restrict__ w2, uint4 *__restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= STEPS * T2 * 64) return;
    const int lane = idx & 63, su = idx >> 6, u = su % T2, s = su / T2;
    const int r = lane & 31, h = lane >> 5, row = 32 * u + r;
    float x[8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float4 v = row < H2 ? *reinterpret_cast<const float4 *>(w2 + (size_t)row * H1 + 16 * s + 8 * g + 4 * h)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        x[4 * g] = v.x; x[4 * g + 1] = v.y; x[4 * g + 2] = v.z; x[4 * g + 3] = v.w;
    }
    uint32_t ph[4], pm[4], pl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2(x[2 * q], x[2 * q + 1], ph[q], pm[q], pl[q]);
    uint4 *dst = packed + ((size_t)s * PIECES + u * 3) * 64 + lane;           // pieces 30, 31 of a step stay zero
    dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    dst[64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
    dst[128] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}

template <bool CRITIC>
__global__ __launch_bounds__(256, 1) void k_mlp_split(const int n, const float *__restrict__ obs,
                                                      const float *__restrict__ action, const Weights W,
                                                      const uint4 *__restrict__ w2p, float *__restrict__ out,
                                                      const ActArgs act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint4 *ring = reinterpret_cast<const uint4 *>(lds_raw);             // 3 x one k16 step of packed fc2
    float *p1_s = reinterpret_cast<float *>(lds_raw + RING_BYTES);            // b1 | g1 | be1          [3][416]
    float *p2_s = p1_s + 3 * H1P;                                             // b2 | g2 | be2 | w3 | wa | ba [6][320]
    float *w1_s = p2_s + 6 * H2P;                                             // raw fc1 [400][23]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * ROWS + wave * WROWS + r;                     // this lane's env (lanes r and r+32 share it)

    // The ring is filled by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = one piece per wave-instruction, no
    // staging registers).  The statements are inline asm, so hipcc neither counts them nor drains them at a barrier:
    // each wave retires its own pieces with a counted s_waitcnt vmcnt before the barrier that precedes their first read.
    const unsigned lane_off = lane * 16;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_raw;
    auto chunk_issue_piece = [&](const int s, const int i) {                  // piece i (0..7) of this wave's share of step s
        const unsigned char *src = reinterpret_cast<const unsigned char *>(w2p) + (size_t)s * CHUNK_BYTES + wave * 8192 + i * 1024;
        const unsigned dst = lds_base + (s % RING) * CHUNK_BYTES + wave * 8192 + i * 1024;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(dst) : "memory");
    };

    NSTAMP(0);
    // ---- prologue: k16 steps 0 and 1 of fc2 start moving; raw fc1 and the per-neuron vectors (zero-padded) -> LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) chunk_issue_piece(i >> 3, i & 7);
    {
        const float4 *src = reinterpret_cast<const float4 *>(W.w1);           // 9200 floats = 2300 float4
        float4 *dst = reinterpret_cast<float4 *>(w1_s);
        for (int i = tid; i < H1 * IN / 4; i += 256) dst[i] = src[i];
        for (int i = tid; i < H1P; i += 256) {
            const bool real = i < H1;
            p1_s[i] = real ? W.b1[i] : 0.f; p1_s[H1P + i] = real ? W.g1[i] : 0.f; p1_s[2 * H1P + i] = real ? W.be1[i] : 0.f;
        }
        for (int i = tid; i < H2P; i += 256) {
            const bool real = i < H2;
            p2_s[i] = real ? W.b2[i] : 0.f; p2_s[H2P + i] = real ? W.g2[i] : 0.f; p2_s[2 * H2P + i] = real ? W.be2[i] : 0.f;
            p2_s[3 * H2P + i] = real ? W.w3[i] : 0.f;
            if (CRITIC) { p2_s[4 * H2P + i] = real ? W.wa[i] : 0.f; p2_s[5 * H2P + i] = real ? W.ba[i] : 0.f; }
        }
    }
    float bx[KQ];                                                             // obs^T as the B operand: k = 2q + h
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int k = 2 * q + h;
        bx[q] = (k < IN && row < n) ? obs[(size_t)row * IN + k] : 0.f;
    }
    __syncthreads();
    NSTAMP(1);

    // ---- layer 1 on the exact f32 MFMA: acc1[t][v] = pre-activation of neuron 32t + 8(v>>2) + 4h + (v&3), env `row`
    f32x16 acc1[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) {
#pragma unroll
        for (int v = 0; v < 16; ++v) acc1[t][v] = 0.f;
        const int m = 32 * t + r;
        float a[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int k = 2 * q + h;
            a[q] = (k < IN && m < H1) ? w1_s[m * IN + k] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bx[q], acc1[t], 0, 0, 0);
    }
    NSTAMP(2);
    // bias, LayerNorm(400) (biased variance, eps 1e-5), ReLU -- in place; tile 12 holds neurons 384..399 in v < 8
    {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int i = 0; i < (t == T1 - 1 ? 2 : 4); ++i) {
                const float4 b = *reinterpret_cast<const float4 *>(p1_s + 32 * t + 8 * i + 4 * h);
                acc1[t][4 * i] += b.x; acc1[t][4 * i + 1] += b.y; acc1[t][4 * i + 2] += b.z; acc1[t][4 * i + 3] += b.w;
                s += (acc1[t][4 * i] + acc1[t][4 * i + 1]) + (acc1[t][4 * i + 2] + acc1[t][4 * i + 3]);
            }
        const float mean = (s + __shfl_xor(s, 32)) * (1.f / H1);
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int v = 0; v < (t == T1 - 1 ? 8 : 16); ++v) { const float d = acc1[t][v] - mean; ss = fmaf(d, d, ss); }
        const float rstd = rsqrtf((ss + __shfl_xor(ss, 32)) * (1.f / H1) + 1e-5f);
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int i = 0; i < (t == T1 - 1 ? 2 : 4); ++i) {
                const float4 g = *reinterpret_cast<const float4 *>(p1_s + H1P + 32 * t + 8 * i + 4 * h);
                const float4 be = *reinterpret_cast<const float4 *>(p1_s + 2 * H1P + 32 * t + 8 * i + 4 * h);
                acc1[t][4 * i] = fmaxf(fmaf((acc1[t][4 * i] - mean) * rstd, g.x, be.x), 0.f);
                acc1[t][4 * i + 1] = fmaxf(fmaf((acc1[t][4 * i + 1] - mean) * rstd, g.y, be.y), 0.f);
                acc1[t][4 * i + 2] = fmaxf(fmaf((acc1[t][4 * i + 2] - mean) * rstd, g.z, be.z), 0.f);
                acc1[t][4 * i + 3] = fmaxf(fmaf((acc1[t][4 * i + 3] - mean) * rstd, g.w, be.w), 0.f);
            }
    }

    NSTAMP(3);
    // ---- layer 2: 25 k16 steps x 10 neuron tiles x 6 bf16 MFMAs; the h1 registers are the B operand
    f32x16 acc2[T2];
#pragma unroll
    for (int u = 0; u < T2; ++u)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc2[u][v] = 0.f;
    // Hand-pipelined issue order, pinned with sched_barrier(0) between every pair of instructions groups: each of the
    // six MFMAs of a tile is followed by one small job that issues while the matrix pipe is busy (an MFMA holds the
    // wave's issue port for 8 of its 32 cycles) -- the three fragment reads of the NEXT tile, then in tiles 0..3 one
    // pair of the next step's B-operand split, in tiles 5..9 the LDS-DMA of step s + 2.  One barrier per step, in the
    // middle (before tile 5): it publishes step s + 1 (every wave has waited for its own pieces) and tells that every
    // wave is past step s - 1, whose slot the DMA of step s + 2 overwrites; tile 0 of step s + 1 can then be
    // prefetched during tile 9 of step s, so no fragment read is ever waited for right after a barrier.
#define SB __builtin_amdgcn_sched_barrier(0)
    uint32_t cb[3][4], nb[3][4];                           // B fragments (hi, mid, lo) of this step and the next
#pragma unroll
    for (int q = 0; q < 4; ++q) split2(acc1[0][2 * q], acc1[0][2 * q + 1], cb[0][q], cb[1][q], cb[2][q]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // steps 0 and 1 (this wave's pieces)
    __syncthreads();
    uint4 a[2][3];
    a[0][0] = ring[lane]; a[0][1] = ring[64 + lane]; a[0][2] = ring[128 + lane];
    SB;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const uint4 vh = make_uint4(cb[0][0], cb[0][1], cb[0][2], cb[0][3]), vm = make_uint4(cb[1][0], cb[1][1], cb[1][2], cb[1][3]),
                    vl = make_uint4(cb[2][0], cb[2][1], cb[2][2], cb[2][3]);
        const uint4 *slot = ring + (s % RING) * CHUNK_U4 + lane;
        const uint4 *nslot = ring + ((s + 1) % RING) * CHUNK_U4 + lane;
#pragma unroll
        for (int u = 0; u < T2; ++u) {
            if (u == T2 / 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                SB;
            }
            const bool more = u + 1 < T2 || s + 1 < STEPS;                    // is there a next tile to prefetch
            const uint4 *nx = u + 1 < T2 ? slot + (u + 1) * 3 * 64 : nslot;
            const uint4 ah = a[u & 1][0], am = a[u & 1][1], al = a[u & 1][2];
            acc2[u] = mfma_bf16(al, vh, acc2[u]); SB;      // small terms first
            if (more) a[(u + 1) & 1][0] = nx[0];
            SB;
            acc2[u] = mfma_bf16(ah, vl, acc2[u]); SB;
            if (more) a[(u + 1) & 1][1] = nx[64];
            SB;
            acc2[u] = mfma_bf16(am, vm, acc2[u]); SB;
            if (more) a[(u + 1) & 1][2] = nx[128];
            SB;
            acc2[u] = mfma_bf16(am, vh, acc2[u]); SB;
            if (u < 4 && s + 1 < STEPS)
                split2(acc1[(s + 1) >> 1][8 * ((s + 1) & 1) + 2 * u], acc1[(s + 1) >> 1][8 * ((s + 1) & 1) + 2 * u + 1],
                       nb[0][u], nb[1][u], nb[2][u]);
            if (u >= 5 && s + 2 < STEPS) chunk_issue_piece(s + 2, u < 8 ? 2 * (u - 5) : u - 2);
            SB;
            acc2[u] = mfma_bf16(ah, vm, acc2[u]); SB;
            if (u >= 5 && u < 8 && s + 2 < STEPS) chunk_issue_piece(s + 2, 2 * (u - 5) + 1);
            SB;
            acc2[u] = mfma_bf16(ah, vh, acc2[u]); SB;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) cb[p][q] = nb[p][q];
    }
#undef SB

    NSTAMP(4);
    // ---- epilogue: bias, LayerNorm(300), (critic: + action_value(a)), ReLU, head; acc2[u][v] is neuron
    // 32u + 8(v>>2) + 4h + (v&3); 300 = 9*32 + 12, so groups of four are all real or all padding
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < T2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m0 = 32 * u + 8 * i + 4 * h;
            const float4 b = *reinterpret_cast<const float4 *>(p2_s + m0);    // zero beyond 300 (so are the products)
            acc2[u][4 * i] += b.x; acc2[u][4 * i + 1] += b.y; acc2[u][4 * i + 2] += b.z; acc2[u][4 * i + 3] += b.w;
            s2 += (acc2[u][4 * i] + acc2[u][4 * i + 1]) + (acc2[u][4 * i + 2] + acc2[u][4 * i + 3]);
        }
    const float mean2 = (s2 + __shfl_xor(s2, 32)) * (1.f / H2);
    float ss2 = 0.f;
#pragma unroll
    for (int u = 0; u < T2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool real = 32 * u + 8 * i + 4 * h < H2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = real ? acc2[u][4 * i + j] - mean2 : 0.f; ss2 = fmaf(d, d, ss2); }
        }
    const float rstd2 = rsqrtf((ss2 + __shfl_xor(ss2, 32)) * (1.f / H2) + 1e-5f);
    const float av = (CRITIC && row < n) ? action[row] : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int u = 0; u < T2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m0 = 32 * u + 8 * i + 4 * h;
            const float4 g = *reinterpret_cast<const float4 *>(p2_s + H2P + m0);
            const float4 be = *reinterpret_cast<const float4 *>(p2_s + 2 * H2P + m0);
            const float4 w3 = *reinterpret_cast<const float4 *>(p2_s + 3 * H2P + m0);
            float y0 = fmaf((acc2[u][4 * i] - mean2) * rstd2, g.x, be.x), y1 = fmaf((acc2[u][4 * i + 1] - mean2) * rstd2, g.y, be.y);
            float y2 = fmaf((acc2[u][4 * i + 2] - mean2) * rstd2, g.z, be.z), y3 = fmaf((acc2[u][4 * i + 3] - mean2) * rstd2, g.w, be.w);
            if (CRITIC) {
                const float4 wa = *reinterpret_cast<const float4 *>(p2_s + 4 * H2P + m0);
                const float4 ba = *reinterpret_cast<const float4 *>(p2_s + 5 * H2P + m0);
                y0 += fmaf(av, wa.x, ba.x); y1 += fmaf(av, wa.y, ba.y); y2 += fmaf(av, wa.z, ba.z); y3 += fmaf(av, wa.w, ba.w);
            }
            dot = fmaf(fmaxf(y0, 0.f), w3.x, dot); dot = fmaf(fmaxf(y1, 0.f), w3.y, dot);      // w3 = 0 on padding
            dot = fmaf(fmaxf(y2, 0.f), w3.z, dot); dot = fmaf(fmaxf(y3, 0.f), w3.w, dot);
        }
    const float v = dot + __shfl_xor(dot, 32) + W.b3[0];
    if (h == 0 && row < n) finish_row<CRITIC>(row, v, out, act);
    NSTAMP(5);
}

}  // namespace

#ifdef TT_STAMPS
int split_debug_stamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_nstamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
#endif

size_t split_ws_bytes() { return (size_t)STEPS * CHUNK_BYTES; }             // 819,200 B

int split_pack(const tt_mlp_weights *w, void *ws, hipStream_t stream) {
    hipLaunchKernelGGL(k_split_pack, dim3((STEPS * T2 * 64 + 255) / 256), dim3(256), 0, stream, w->w2,
                       reinterpret_cast<uint4 *>(ws));
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

template <bool CRITIC>
static int launch_split(int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                        const ActArgs &act, hipStream_t stream) {
    static bool attr[64] = {};      // per device: a function attribute set on one device says nothing about another
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TT_EHIP;
    if (!attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_split<CRITIC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return TT_EHIP;
        attr[dev] = true;
    }
    hipLaunchKernelGGL((k_mlp_split<CRITIC>), dim3((n + ROWS - 1) / ROWS), dim3(256), LDS_BYTES, stream, n, obs, action,
                       to_weights(w), reinterpret_cast<const uint4 *>(w->split_ws), out, act);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int split_forward(bool critic, int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                  const ActArgs &act, hipStream_t stream) {
    return critic ? launch_split<true>(n, obs, action, w, out, act, stream)
                  : launch_split<false>(n, obs, action, w, out, act, stream);
}

}  // namespace ttnet
