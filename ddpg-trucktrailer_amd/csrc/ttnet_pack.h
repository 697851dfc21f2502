// ttnet_pack.h -- the split-f16 policy IMAGE of a network (csrc/ttnet_split.hip reads it; DDPG/networks.py:98-147 the
// modules it is made from): geometry constants, the f32 -> two-f16-pieces split, and the body of the pack launch.  Shared by
// csrc/ttnet_split.hip (k_split_pack, k_pack_and_sample, the forward itself) and csrc/ttlearn.hip, whose critic-backward launch
// can carry the pack of a vector step's image on workgroups of its own (tt_mlp_backward_rows_pair: tt_image_job).
#pragma once
#include "ttnet_common.h"

namespace ttnet {

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


}  // namespace ttnet
