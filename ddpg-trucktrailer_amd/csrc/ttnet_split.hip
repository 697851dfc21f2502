// ttnet_split.hip -- the N-env forward of the reference's actor / critic MLPs (DDPG/networks.py:55-68, 138-147) on the
// 16-bit MFMA of gfx950 at f32 accuracy ("split-f16").
//
// Why: the f32-input MFMA runs at 1/16 of the f16/bf16 rate (155 TFLOP/s vs ~2.5 PFLOP/s dense), and fc2 (400 -> 300)
// is 93 % of the network's arithmetic.  An f32 number x (24 significand bits) is, to within its own rounding error, the
// sum of TWO f16 numbers when both conversions round to nearest: h = rn16(x) leaves |x - h| <= 2^-12 |x| and
// m = rn16(x - h) leaves |x - h - m| <= 2^-24 |x| (11 + 1 + 11 + 1 bits; the remainder x - h is exact in f32).  A
// product of two f16 numbers is exact in f32 (22 bits), so
//     x*w = h*h' + (h*m' + m*h') + [m*m' + (x - h - m)*w + x*(w - h' - m')]
// where every term of the bracket is below 2^-24 |x*w|: three f16 MFMAs with f32 accumulation reproduce the f32 product
// to within ~3 of its own rounding errors, at 3/16 of the f32 MFMA's cycles (the round-1 kernel used three bf16 pieces
// and six MFMAs per product block).  f16 has 5 exponent bits, so both operands are scaled by powers of two (exact) to
// keep their `m` pieces normal: activations by 2^4, weights by 2^6; the accumulator is scaled back by 2^-10 (exact)
// where the bias is added.  Ranges: |activation| < 4094, |weight| < 1023 -- beyond them a piece becomes inf and the row's
// output NaN (loud, not silently wrong); pieces below the normal range carry an absolute error <= 2^-25 / scale.
//
// Orientation: everything is computed TRANSPOSED (neurons on the MFMA's M axis, envs on N), with
// v_mfma_f32_32x32x16_f16.  Its 32x32 result has the env on the lane (lane & 31) and 16 neurons in the lane's
// registers, which is exactly the B-operand layout of the next product that sums over neurons: the layer-1 result
// goes through LayerNorm + ReLU in registers and feeds layer 2 with no LDS round trip and no lane movement
// (the k order inside a step is permuted; the packed weights are laid out to match).  Both LayerNorms and the
// 300 -> 1 head reduce over registers plus ONE cross-lane exchange (lane ^ 32).
//
//   workgroup = 4 waves = 128 envs (32 per wave); per wave: h1 = 13 tiles x 16 registers (f32), layer-2
//   accumulators 10 tiles x 16 registers; fc2 is streamed once per workgroup, as pre-split f16 planes in MFMA
//   fragment order, through a 7-slot LDS ring shared by the four waves (20 KB per k16 step), filled by LDS-DMA five
//   steps ahead of its use.
//   Layer 1 (K = 23 + the bias as a 24th input that is always 1) runs on the same three-product scheme from pre-split
//   fc1 fragments, also brought into LDS by DMA.
#include "ttnet_common.h"

namespace ttnet {
namespace {

#ifdef TT_STAMPS   // diagnostic build only: clocks of thread 0 of every workgroup at the phase boundaries
__device__ unsigned long long g_nstamps[16];
__device__ unsigned long long g_bstamps[4096 * 8];      // [block][phase] wall clock (100 MHz); [block][7] = XCC id
#define NSTAMP(i) do { if (threadIdx.x == 0) { const unsigned long long w_ = wall_clock64();                                \
        if (blockIdx.x < 4096) { g_bstamps[blockIdx.x * 8 + (i)] = w_;                                                      \
            if ((i) == 0) g_bstamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); }     \
        if (blockIdx.x == 0) { g_nstamps[i] = w_; g_nstamps[8 + i] = clock64(); } } } while (0)
#else
#define NSTAMP(i) do { } while (0)
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr int ROWS = 128, WROWS = 32;       // envs per workgroup / per wave
constexpr int T1 = 13, H1P = 32 * T1;       // layer-1 neuron tiles (416 >= 400)
constexpr int T2 = 10, H2P = 32 * T2;       // layer-2 neuron tiles (320 >= 300)
#ifdef TT_DBG_STEPS                         // timing experiments only (wrong results): fewer k16 steps of layer 2
constexpr int STEPS = TT_DBG_STEPS;
#else
constexpr int STEPS = H1 / 16;              // 25 k16 steps of layer 2
#endif
constexpr int STEPS_FULL = H1 / 16;         // the h1 pieces always cover all 25 steps
constexpr int S1 = 2;                       // k16 steps of layer 1: inputs 0..22 = observation, 23 = 1 (bias), 24..31 = 0
constexpr int PIECES = 2 * T2;              // one k16 step of packed fc2 = 10 tiles x 2 planes of 64 lanes x 16 B: 5 per wave
constexpr int CHUNK_U4 = PIECES * 64;
constexpr int CHUNK_BYTES = CHUNK_U4 * 16;  // 20,480
constexpr int RING = 7;                     // an LDS-DMA piece lands ~1 us after its issue, a step is consumed in ~0.5 us: the
                                            // stream runs RING - 2 = 5 steps ahead of the step being consumed
constexpr int AHEAD = RING - 2;
constexpr int RING_BYTES = RING * CHUNK_BYTES;                        // 143,360
constexpr int W1_PIECES = S1 * T1 * 2;      // 52 pieces of packed fc1: 13 per wave
constexpr int W1_BYTES = W1_PIECES * 1024;  // 53,248: lives in ring slots 2..4 until layer 1 is done
constexpr int W1_OFF = 2 * CHUNK_BYTES;
static_assert(W1_OFF + W1_BYTES <= AHEAD * CHUNK_BYTES, "packed fc1 must fit in the ring slots that are filled after layer 1");
constexpr int VEC_FLOATS = 2 * H1P + 6 * H2P;                         // g1' | be1' | b2 | g2 | be2 | w3 | wa | ba = 2,752
constexpr int VEC_PIECES = 12;              // 11,008 B padded to 12 KB: 3 DMA pieces per wave
constexpr int VEC_BYTES = VEC_PIECES * 1024;
constexpr int LDS_BYTES = RING_BYTES + VEC_BYTES;                     // 155,648
constexpr int WS_W2 = 0, WS_W1 = STEPS * CHUNK_BYTES, WS_VEC = WS_W1 + W1_BYTES, WS_BYTES = WS_VEC + VEC_BYTES;   // 577,536

constexpr float SX = 16.f, SW = 64.f;       // operand scales (powers of two: exact)
constexpr float UNSCALE = 1.f / (SX * SW);

// x0, x1 -> their two f16 pieces, packed as (x0 piece | x1 piece << 16); both conversions round to nearest even.
// v_fma_mix_f32 forms the exact remainder x - (float)h straight from the packed half.
__device__ __forceinline__ void split2(const float x0, const float x1, uint32_t &ph, uint32_t &pm) {
    h2 h;
    h[0] = (_Float16)x0; h[1] = (_Float16)x1;                       // one v_cvt_pk_f16_f32
    ph = __builtin_bit_cast(uint32_t, h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(ph), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(ph), "v"(x1));
    h2 m;
    m[0] = (_Float16)r0; m[1] = (_Float16)r1;
    pm = __builtin_bit_cast(uint32_t, m);
}

// Packed f32 arithmetic (v_pk_add/mul/fma_f32: two lanes' worth of f32 per instruction at the same issue cost): the
// vector phases of the forward are issue-bound on one wave per SIMD, and their operands (accumulator registers, float4s of
// per-neuron vectors) already sit in even-aligned register pairs.  Component-wise IEEE fma: the same numbers as scalar code.
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 pk(const float a, const float b) { return f32x2{a, b}; }
__device__ __forceinline__ f32x2 pk_fma(const f32x2 a, const f32x2 b, const f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f32x16 mfma_f16(const uint4 a, const uint4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Workspace image of one network (re-made by every call that uses it: never stale):
//   fc2 [300,400] -> w2p[s][u][plane][lane] (16 B each): lane (r = lane & 31, h = lane >> 5) of neuron tile u holds, for
//     k16 step s, the weights of neuron 32u + r at inputs k(j) = 16 s + 8 (j >> 2) + 4 h + (j & 3), j = 0..7 -- the order
//     in which a 32x32 accumulator tile presents its rows when it is used as the other operand; plane 0 = h, 1 = m;
//   fc1 [400,23] + bias -> w1p[s][t][plane][lane]: neuron 32t + r at inputs 16 s + 8 h + j (natural order; input 23
//     carries the bias, the observation operand carries a 1 there);
//   vec: g1*SX | be1*SX [2][416], b2 | g2 | be2 | w3 | wa | ba [6][320], zero beyond the real neurons; then b3.
__device__ __forceinline__ void split_pack_body(const Weights &W, const bool critic, unsigned char *__restrict__ ws,
                                                unsigned char *__restrict__ ws_alt, long long *__restrict__ bump,
                                                const RingCursor &cur, const int idx) {
    // two images (tt_mlp_weights.split_ws_alt): the one of the parity of the step this launch opens, so that the launch may
    // run beside a forward of the previous step that still reads the other
    if (ws_alt && cur.k_dev && (*cur.k_dev & 1)) ws = ws_alt;
    if (idx == 0) {
        if (bump) *bump += 1;              // optional step counter of a pipelined loop (read by LATER launches only)
        write_cursor(cur);                 // ring slots of the step this launch opens (nothing advances k_dev right now)
    }
    constexpr int N2 = STEPS * T2 * 64, N1 = S1 * T1 * 64;
    if (idx < N2 + N1) {
        const bool l2 = idx < N2;
        const int q = l2 ? idx : idx - N2;
        const int lane = q & 63, su = q >> 6, nt = l2 ? T2 : T1, u = su % nt, s = su / nt;
        const int r = lane & 31, h = lane >> 5, row = 32 * u + r;
        float x[8];
        if (l2) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 v = row < H2 ? *reinterpret_cast<const float4 *>(W.w2 + (size_t)row * H1 + 16 * s + 8 * g + 4 * h)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                x[4 * g] = v.x; x[4 * g + 1] = v.y; x[4 * g + 2] = v.z; x[4 * g + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * h + j;
                x[j] = row < H1 ? (k < IN ? W.w1[row * IN + k] : (k == IN ? W.b1[row] : 0.f)) : 0.f;
            }
        }
        uint32_t ph[4], pm[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) split2(x[2 * p] * SW, x[2 * p + 1] * SW, ph[p], pm[p]);
        uint4 *dst = reinterpret_cast<uint4 *>(ws + (l2 ? WS_W2 : WS_W1)) + ((size_t)su * 2) * 64 + lane;
        dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        dst[64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        return;
    }
    const int i = idx - (N2 + N1);
    if (i >= VEC_BYTES / 4) return;
    float *vec = reinterpret_cast<float *>(ws + WS_VEC);
    float v = 0.f;
    if (i < 2 * H1P) {
        const int a = i / H1P, m = i % H1P;
        if (m < H1) v = (a == 0 ? W.g1[m] : W.be1[m]) * SX;
    } else if (i < VEC_FLOATS) {
        const int a = (i - 2 * H1P) / H2P, m = (i - 2 * H1P) % H2P;
        if (m < H2) {
            const float *src[6] = {W.b2, W.g2, W.be2, W.w3, W.wa, W.ba};
            if (a < 4 || critic) v = src[a][m];
        }
    } else if (i == VEC_FLOATS) {
        v = W.b3[0];                       // the head's bias travels with the image: the forward reads NOTHING else of the net
    }
    vec[i] = v;
}
constexpr int PACK_THREADS = (STEPS * T2 + S1 * T1) * 64 + VEC_BYTES / 4;
constexpr int PACK_BLOCKS = (PACK_THREADS + 255) / 256;

__global__ __launch_bounds__(256) void k_split_pack(const Weights W, const bool critic, unsigned char *__restrict__ ws,
                                                    unsigned char *__restrict__ ws_alt, long long *__restrict__ bump,
                                                    const RingCursor cur) {
    split_pack_body(W, critic, ws, ws_alt, bump, cur, blockIdx.x * 256 + threadIdx.x);
}

// What opens a pipelined vector step, in ONE launch (two small kernels would each cost their ~4 us of launch and a
// dependency gap on the loop's critical path): the policy's image from the actor's current weights, and the first batch of
// this step's learn() from the replay ring (four sampled transitions per 256-thread workgroup).
__global__ __launch_bounds__(256) void k_pack_and_sample(const Weights W, const bool critic, unsigned char *__restrict__ ws,
                                                         unsigned char *__restrict__ ws_alt, const RingSample R,
                                                         const RingCursor cur) {
    if ((int)blockIdx.x < PACK_BLOCKS) {
        split_pack_body(W, critic, ws, ws_alt, nullptr, cur, blockIdx.x * 256 + threadIdx.x);
        return;
    }
    const int b = ((int)blockIdx.x - PACK_BLOCKS) * 4 + (threadIdx.x >> 6);
    if (b < R.batch) ring_sample_row(R, b, threadIdx.x & 63);
}

template <bool CRITIC>
__global__ __launch_bounds__(256, 1) void k_mlp_split(const int n, const int tile0, const float *__restrict__ obs,
                                                      const float *__restrict__ action,
                                                      const unsigned char *__restrict__ ws,
                                                      const unsigned char *__restrict__ ws_alt, float *__restrict__ out,
                                                      const ActArgs act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint4 *ring = reinterpret_cast<const uint4 *>(lds_raw);             // RING x one k16 step of packed fc2
    const uint4 *w1_s = reinterpret_cast<const uint4 *>(lds_raw + W1_OFF);    // packed fc1 [2][13][2][64] (in ring slots 2..4)
    const float *p1_s = reinterpret_cast<const float *>(lds_raw + RING_BYTES);              // g1' | be1'       [2][416]
    const float *p2_s = p1_s + 2 * H1P;                                       // b2 | g2 | be2 | w3 | wa | ba [6][320]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // this lane's view of the per-neuron vectors, its address held in a register: the region lies beyond the 64 KB a
    // ds_read's immediate offset reaches from address 0, and left to itself the compiler forms every address with an add
    using lds_f = const __attribute__((address_space(3))) float;
    using v4 = __attribute__((ext_vector_type(4))) float;
    auto lds4 = [](lds_f *q) -> float4 {
        const v4 t = *(const __attribute__((address_space(3))) v4 *)q;
        return make_float4(t[0], t[1], t[2], t[3]);
    };
    lds_f *pv1 = (lds_f *)(p1_s + 4 * h);
    asm volatile("" : "+v"(pv1));
    lds_f *pv2 = pv1 + 2 * H1P;
    // One 128-env tile per workgroup (a loop over tiles inside the kernel costs ~400 spilled registers: the compiler hoists
    // the tile-invariant DMA addresses); a caller that wants CUs left free launches the tiles in several grids (tile0).
    const int tile = tile0 + blockIdx.x;
    // ring addressing: the image of the running step's parity (the other one may be being written for the next step), and
    // the first workgroup leaves the running step's cursor where the env step reads it
    const bool odd = act.cursor && (*act.step_dev & 1);
    const unsigned char *wsl = (odd && ws_alt) ? ws_alt : ws;
    if (act.cursor && tile == 0 && tid < 4) act.cursor[tid] = cursor_of(act)[tid];
    const int row = tile * ROWS + wave * WROWS + r;                           // this lane's env (lanes r and r+32 share it)

    // LDS is filled by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = one 1 KB piece per wave-instruction, no
    // staging registers).  The statements are inline asm, so hipcc neither counts them nor drains them at a barrier:
    // each wave retires its own pieces with an s_waitcnt vmcnt before the barrier that precedes their first read.
    const unsigned lane_off = lane * 16;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_raw;
    auto dma_piece = [&](const unsigned char *src, const unsigned dst) {      // src, dst wave-uniform; + lane * 16 each
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(dst) : "memory");
    };
    const unsigned char *wsl_cur = wsl;
    auto chunk_issue_piece = [&](const int s, const int i) {                  // piece i (0..4) of this wave's share of step s
        dma_piece(wsl_cur + WS_W2 + (size_t)s * CHUNK_BYTES + wave * 5120 + i * 1024,
                  lds_base + (s % RING) * CHUNK_BYTES + wave * 5120 + i * 1024);
    };

    NSTAMP(0);
    // ---- prologue: this lane's observation features first (ordinary loads: the compiler waits for them on its own
    // count), then packed fc1 + the per-neuron vectors, then k16 steps 0 and 1 of fc2 by DMA
    float xo[S1][8];                                                          // obs^T as the B operand: input 16 s + 8 h + j
    // unconditional loads from clamped (always valid) addresses, selected afterwards: 16 loads in flight at once
    // instead of 16 exec-masked round trips
    const float *orow = resolve_obs(act, obs) + (size_t)(row < n ? row : n - 1) * IN;
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            xo[s][j] = orow[k < IN ? k : IN - 1];
        }
#pragma unroll
    for (int i = 0; i < W1_PIECES / 4; ++i)
        dma_piece(wsl + WS_W1 + (wave * (W1_PIECES / 4) + i) * 1024, lds_base + W1_OFF + (wave * (W1_PIECES / 4) + i) * 1024);
#pragma unroll
    for (int i = 0; i < VEC_PIECES / 4; ++i)
        dma_piece(wsl + WS_VEC + (wave * (VEC_PIECES / 4) + i) * 1024,
                  lds_base + RING_BYTES + (wave * (VEC_PIECES / 4) + i) * 1024);
#pragma unroll
    for (int i = 0; i < 10; ++i) chunk_issue_piece(i / 5, i % 5);
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            xo[s][j] = k < IN ? (row < n ? xo[s][j] : 0.f) : (k == IN ? 1.f : 0.f);
        }
    uint4 xb[S1][2];                                                          // the observation's h and m pieces
#pragma unroll
    for (int s = 0; s < S1; ++s) {
        uint32_t ph[4], pm[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) split2(xo[s][2 * p] * SX, xo[s][2 * p + 1] * SX, ph[p], pm[p]);
        xb[s][0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        xb[s][1] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
    }
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                         // fc1 + vectors landed; fc2 steps 0, 1 may still fly
    __builtin_amdgcn_s_barrier();
    NSTAMP(1);

    // ---- layer 1: acc1[t][v] = SX*SW * (pre-activation + bias) of neuron 32t + 8(v>>2) + 4h + (v&3), env `row`
    f32x16 acc1[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) {
#pragma unroll
        for (int v = 0; v < 16; ++v) acc1[t][v] = 0.f;
#pragma unroll
        for (int s = 0; s < S1; ++s) {
            const uint4 ah = w1_s[((s * T1 + t) * 2) * 64 + lane], am = w1_s[((s * T1 + t) * 2 + 1) * 64 + lane];
            acc1[t] = mfma_f16(am, xb[s][0], acc1[t]);                        // small terms first
            acc1[t] = mfma_f16(ah, xb[s][1], acc1[t]);
            acc1[t] = mfma_f16(ah, xb[s][0], acc1[t]);
        }
    }
    NSTAMP(2);
    // every wave is done with packed fc1: its slots now take k16 steps 2..5 of fc2 (they land during LayerNorm 1)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 10; i < 5 * (AHEAD + 1); ++i) chunk_issue_piece(i / 5, i % 5);
    uint32_t hb[STEPS_FULL][4], mb[STEPS_FULL][4];
    // LayerNorm(400) (biased variance, eps 1e-5) on the SCALED pre-activations: mean and deviations scale with them,
    // 1/sigma absorbs the scale; then gamma*SX, beta*SX and ReLU give the layer-2 operand already scaled by SX.
    // Tile 12 holds neurons 384..399 in v < 8.
    {
        f32x2 sp = pk(0.f, 0.f);
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int v = 0; v < (t == T1 - 1 ? 8 : 16); v += 4)
                sp += pk(acc1[t][v], acc1[t][v + 1]) + pk(acc1[t][v + 2], acc1[t][v + 3]);
        const float s = sp[0] + sp[1];
        const float mean = (s + __shfl_xor(s, 32)) * (1.f / H1);
        const f32x2 nmean = pk(-mean, -mean);
        f32x2 ssp = pk(0.f, 0.f);
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int v = 0; v < (t == T1 - 1 ? 8 : 16); v += 2) {
                const f32x2 d = pk(acc1[t][v], acc1[t][v + 1]) + nmean;
                acc1[t][v] = d[0]; acc1[t][v + 1] = d[1];
                ssp = pk_fma(d, d, ssp);
            }
        const float ss = ssp[0] + ssp[1];
        const float var = (ss + __shfl_xor(ss, 32)) * (1.f / H1) * (UNSCALE * UNSCALE);
        const float rstd = rsqrtf(var + 1e-5f) * UNSCALE;
        const f32x2 rstdp = pk(rstd, rstd);
        // gamma*SX, beta*SX, ReLU, then straight into the two f16 pieces layer 2 multiplies with: hb[s] / mb[s] are the
        // B fragments (h, m planes) of k16 step s -- element j of lane half h is neuron 16 s + 8 (j >> 2) + 4 h + (j & 3),
        // i.e. registers 8 (s & 1) .. + 7 of tile s >> 1, the order the packed fc2 fragments are laid out in
        // The per-neuron vectors come from LDS a whole tile AHEAD of their use (8 ds_read_b128, pinned above the tile's
        // arithmetic): a lone wave hides no latency, and read-then-use cost one exposed LDS round trip per group of four.
        float4 gq[2][4], bq[2][4];
        auto ld1 = [&](const int t, float4 (&g)[4], float4 (&be)[4]) {
#pragma unroll
            for (int i = 0; i < (t == T1 - 1 ? 2 : 4); ++i) {
                g[i] = lds4(pv1 + 32 * t + 8 * i);
                be[i] = lds4(pv1 + H1P + 32 * t + 8 * i);
            }
        };
        ld1(0, gq[0], bq[0]);
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            if (t + 1 < T1) ld1(t + 1, gq[(t + 1) & 1], bq[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < (t == T1 - 1 ? 2 : 4); ++i) {
                const float4 g = gq[t & 1][i], be = bq[t & 1][i];
                const f32x2 ya = pk_fma(pk(acc1[t][4 * i], acc1[t][4 * i + 1]) * rstdp, pk(g.x, g.y), pk(be.x, be.y));
                const f32x2 yb = pk_fma(pk(acc1[t][4 * i + 2], acc1[t][4 * i + 3]) * rstdp, pk(g.z, g.w), pk(be.z, be.w));
                const float y0 = fmaxf(ya[0], 0.f), y1 = fmaxf(ya[1], 0.f), y2 = fmaxf(yb[0], 0.f), y3 = fmaxf(yb[1], 0.f);
                const int st = 2 * t + (i >> 1), e = 2 * (i & 1);             // k16 step, first of its two packed registers
                split2(y0, y1, hb[st][e], mb[st][e]);
                split2(y2, y3, hb[st][e + 1], mb[st][e + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    NSTAMP(3);
    // ---- layer 2: 25 k16 steps x 10 neuron tiles x 3 f16 MFMAs; the h1 registers are the B operand
    f32x16 acc2[T2];
#pragma unroll
    for (int u = 0; u < T2; ++u)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc2[u][v] = 0.f;
    // Hand-pipelined issue order, pinned with sched_barrier(0) between every pair of instruction groups: each of the
    // three MFMAs of a tile is followed by one small job that issues while the matrix pipe is busy (an MFMA holds the
    // wave's issue port for 8 of its 32 cycles) -- the two fragment reads of the tile two ahead, and in tiles 5..9 one
    // LDS-DMA piece of step s + 6.  No vector arithmetic is left in the loop: the B operand was split once, in LayerNorm 1.  One barrier per step, in
    // the middle (before tile 5): it publishes step s + 1 (every wave has waited for its own pieces of it with a COUNTED
    // vmcnt that leaves the four younger steps in flight) and tells that every wave is past step s - 1, whose slot the
    // DMA of step s + 6 overwrites.  Fragments are read TWO tiles ahead
    // (tiles 0, 1 of step s + 1 during tiles 8, 9 of step s, after that barrier), so a read has six MFMAs to land.
#define SB __builtin_amdgcn_sched_barrier(0)
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // steps 0 and 1 (this wave's pieces); 2..5 may still fly
    __builtin_amdgcn_s_barrier();
    // A fragments (h, m) of three consecutive tiles: the current one, the next, and the one being read (two ahead);
    // rotated by renaming at the end of every tile (the loops are fully unrolled: no moves)
    uint4 c0h = ring[lane], c0m = ring[64 + lane], c1h = ring[128 + lane], c1m = ring[192 + lane], c2h = c0h, c2m = c0m;
    SB;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const uint4 vh = make_uint4(hb[s][0], hb[s][1], hb[s][2], hb[s][3]), vm = make_uint4(mb[s][0], mb[s][1], mb[s][2], mb[s][3]);
        const uint4 *slot = ring + (s % RING) * CHUNK_U4 + lane;
        const uint4 *nslot = ring + ((s + 1) % RING) * CHUNK_U4 + lane;
#pragma unroll
        for (int u = 0; u < T2; ++u) {
            if (u == T2 / 2) {
                // step s + 1 must have landed; the steps issued after it (up to four) may still be in flight
                switch (STEPS - 2 - s < 4 ? STEPS - 2 - s : 4) {
                    case 4: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                }
#ifndef TT_DBG_NOBAR
                __builtin_amdgcn_s_barrier();
#endif
                SB;
            }
            const bool more = u + 2 < T2 || s + 1 < STEPS;                    // is there a tile two ahead to prefetch
            const uint4 *nx = u + 2 < T2 ? slot + (u + 2) * 2 * 64 : nslot + (u + 2 - T2) * 2 * 64;
            acc2[u] = mfma_f16(c0h, vm, acc2[u]); SB;      // small terms first
            if (more) c2h = nx[0];
            SB;
            acc2[u] = mfma_f16(c0m, vh, acc2[u]); SB;
            if (more) c2m = nx[64];
            SB;
            acc2[u] = mfma_f16(c0h, vh, acc2[u]); SB;
#ifndef TT_DBG_NODMA
            if (u >= 5 && s + RING - 1 < STEPS) chunk_issue_piece(s + RING - 1, u - 5);
#endif
            SB;
            c0h = c1h; c0m = c1m; c1h = c2h; c1m = c2m;
        }
    }
#undef SB

    NSTAMP(4);
    // ---- epilogue: scale back + bias, LayerNorm(300), (critic: + action_value(a)), ReLU, head; acc2[u][v] is neuron
    // 32u + 8(v>>2) + 4h + (v&3); 300 = 9*32 + 12, so groups of four are all real or all padding
    // (the per-neuron vectors are read from LDS one tile ahead of their use, as in LayerNorm 1)
    f32x2 s2p = pk(0.f, 0.f);
    {
        float4 bq[2][4];
        auto ldb = [&](const int u, float4 (&b)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = lds4(pv2 + 32 * u + 8 * i);   // zero beyond 300
        };
        ldb(0, bq[0]);
#pragma unroll
        for (int u = 0; u < T2; ++u) {
            if (u + 1 < T2) ldb(u + 1, bq[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 b = bq[u & 1][i];
                const f32x2 xa = pk_fma(pk(acc2[u][4 * i], acc2[u][4 * i + 1]), pk(UNSCALE, UNSCALE), pk(b.x, b.y));
                const f32x2 xb2 = pk_fma(pk(acc2[u][4 * i + 2], acc2[u][4 * i + 3]), pk(UNSCALE, UNSCALE), pk(b.z, b.w));
                acc2[u][4 * i] = xa[0]; acc2[u][4 * i + 1] = xa[1]; acc2[u][4 * i + 2] = xb2[0]; acc2[u][4 * i + 3] = xb2[1];
                s2p += xa + xb2;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float s2 = s2p[0] + s2p[1];
    const float mean2 = (s2 + __shfl_xor(s2, 32)) * (1.f / H2);
    const f32x2 nmean2 = pk(-mean2, -mean2);
    // deviations in place (the last pass needs nothing else of the pre-activations); padding groups do not count
    f32x2 ss2p = pk(0.f, 0.f);
#pragma unroll
    for (int u = 0; u < T2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool real = 32 * u + 8 * i + 4 * h < H2;
            const f32x2 keep = real ? pk(1.f, 1.f) : pk(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const f32x2 d = pk(acc2[u][4 * i + j], acc2[u][4 * i + j + 1]) + nmean2;
                acc2[u][4 * i + j] = d[0]; acc2[u][4 * i + j + 1] = d[1];
                const f32x2 dk = (32 * u + 8 * i + 4 < H2) ? d : d * keep;      // only the last tiles can hold padding
                ss2p = pk_fma(dk, dk, ss2p);
            }
        }
    const float ss2 = ss2p[0] + ss2p[1];
    const float rstd2 = rsqrtf((ss2 + __shfl_xor(ss2, 32)) * (1.f / H2) + 1e-5f);
    const f32x2 rstd2p = pk(rstd2, rstd2);
    const float av = (CRITIC && row < n) ? action[row] : 0.f;
    f32x2 dotp = pk(0.f, 0.f);
    {
        constexpr int NV = CRITIC ? 5 : 3;                  // gamma2, beta2, w3 (, wa, ba)
        float4 vq[2][4][NV];
        auto ldv = [&](const int u, float4 (&v)[4][NV]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < NV; ++q) v[i][q] = lds4(pv2 + (q + 1) * H2P + 32 * u + 8 * i);
        };
        ldv(0, vq[0]);
#pragma unroll
        for (int u = 0; u < T2; ++u) {
            if (u + 1 < T2) ldv(u + 1, vq[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 g = vq[u & 1][i][0], be = vq[u & 1][i][1], w3 = vq[u & 1][i][2];
                f32x2 ya = pk_fma(pk(acc2[u][4 * i], acc2[u][4 * i + 1]) * rstd2p, pk(g.x, g.y), pk(be.x, be.y));
                f32x2 yb = pk_fma(pk(acc2[u][4 * i + 2], acc2[u][4 * i + 3]) * rstd2p, pk(g.z, g.w), pk(be.z, be.w));
                if (CRITIC) {
                    const float4 wa = vq[u & 1][i][NV - 2], ba = vq[u & 1][i][NV - 1];
                    ya += pk_fma(pk(av, av), pk(wa.x, wa.y), pk(ba.x, ba.y));
                    yb += pk_fma(pk(av, av), pk(wa.z, wa.w), pk(ba.z, ba.w));
                }
                dotp = pk_fma(pk(fmaxf(ya[0], 0.f), fmaxf(ya[1], 0.f)), pk(w3.x, w3.y), dotp);      // w3 = 0 on padding
                dotp = pk_fma(pk(fmaxf(yb[0], 0.f), fmaxf(yb[1], 0.f)), pk(w3.z, w3.w), dotp);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float dot = dotp[0] + dotp[1];
    const float v = dot + __shfl_xor(dot, 32) + p1_s[VEC_FLOATS];
    if (h == 0 && row < n) finish_row<CRITIC>(row, v, out, act);
    NSTAMP(5);
}

}  // namespace

#ifdef TT_STAMPS
int split_debug_stamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_nstamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -3;
}
int split_debug_block_stamps(unsigned long long *out, int nblocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamps), sizeof(unsigned long long) * 8 * (nblocks < 4096 ? nblocks : 4096)) == hipSuccess ? 0 : -3;
}
#endif

size_t split_ws_bytes() { return (size_t)WS_BYTES; }

int split_pack(const tt_mlp_weights *w, bool critic, void *ws, long long *bump, const RingCursor &cur, hipStream_t stream) {
    // (the second image takes part only when the caller packs into the struct's own workspace)
    unsigned char *alt = ws == w->split_ws ? reinterpret_cast<unsigned char *>(w->split_ws_alt) : nullptr;
    hipLaunchKernelGGL(k_split_pack, dim3(PACK_BLOCKS), dim3(256), 0, stream, to_weights(w), critic,
                       reinterpret_cast<unsigned char *>(ws), alt, bump, cur);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int split_pack_and_sample(const tt_mlp_weights *w, bool critic, void *ws, const RingSample &R, const RingCursor &cur,
                          hipStream_t stream) {
    unsigned char *alt = ws == w->split_ws ? reinterpret_cast<unsigned char *>(w->split_ws_alt) : nullptr;
    hipLaunchKernelGGL(k_pack_and_sample, dim3(PACK_BLOCKS + (R.batch + 3) / 4), dim3(256), 0, stream, to_weights(w), critic,
                       reinterpret_cast<unsigned char *>(ws), alt, R, cur);
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
    // one workgroup per CU is resident (LDS).  max_workgroups > 0: the tiles go out in consecutive grids of at most that
    // many workgroups, so that never more than that many CUs are busy with this forward (the rest stay free for launches on
    // other streams); 0: one grid, the hardware takes the tiles in rounds
    const int ntiles = (n + ROWS - 1) / ROWS;
    // capped_grids > 0: only that many grids are capped -- what the cap makes room for is over by then -- and the rest of the
    // tiles go out in one grid over the whole chip
    const int limit = w->max_workgroups > 0 ? w->max_workgroups : ntiles;
    int capped = ntiles;
    if (w->max_workgroups > 0 && w->capped_grids > 0 && (long long)w->capped_grids * limit < ntiles)
        capped = w->capped_grids * limit;
    const int grids = (capped + limit - 1) / limit, cap = (capped + grids - 1) / grids;      // equal shares: 512 tiles, limit 192 -> 171, 171, 170
    for (int t0 = 0; t0 < capped; t0 += cap) {
        const int g = capped - t0 < cap ? capped - t0 : cap;
        hipLaunchKernelGGL((k_mlp_split<CRITIC>), dim3(g), dim3(256), LDS_BYTES, stream, n, t0, obs, action,
                           reinterpret_cast<const unsigned char *>(w->split_ws),
                           reinterpret_cast<const unsigned char *>(w->split_ws_alt), out, act);
    }
    if (capped < ntiles)
        hipLaunchKernelGGL((k_mlp_split<CRITIC>), dim3(ntiles - capped), dim3(256), LDS_BYTES, stream, n, capped, obs, action,
                           reinterpret_cast<const unsigned char *>(w->split_ws),
                           reinterpret_cast<const unsigned char *>(w->split_ws_alt), out, act);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int split_forward(bool critic, int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                  const ActArgs &act, hipStream_t stream) {
    return critic ? launch_split<true>(n, obs, action, w, out, act, stream)
                  : launch_split<false>(n, obs, action, w, out, act, stream);
}

}  // namespace ttnet
