// ttnet.hip -- fused inference of the reference's actor / critic MLPs on MI355X (gfx950), exact f32.
//
//   actor  (DDPG/networks.py:138-147): tanh(mu(relu(bn2(fc2(relu(bn1(fc1(s))))))))
//   critic (DDPG/networks.py:55-68):   q(relu(bn2(fc2(relu(bn1(fc1(s))))) + action_value(a)))
// with fc1 23->400, fc2 400->300, LayerNorm (eps 1e-5, biased variance) -- the shapes of trainv2.py:404-407.
//
// One workgroup (4 waves) pushes a tile of 64 rows through the whole network; each wave owns 16 rows and ALL
// output columns, so both LayerNorms and the final dot product reduce inside the wave.  Matrix products run on
// the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered fma chain); activations never leave the
// CU: layer-1 output goes registers -> LayerNorm/ReLU in registers -> LDS (as the A operand of layer 2), the
// layer-2 accumulators stay in registers through LayerNorm, ReLU, the 300->1 head and tanh.  fc2's weights
// (480 KB, L2-resident) stream through a double-buffered LDS ring in 16-column chunks.
// LDS strides (402 / 26 / 18 words) make every MFMA fragment read conflict-free (18*j mod 32 distinct evens).
//
// The actor kernel can also finish DDPG_agent.choose_action + trainv2.py:516 in its epilogue: OU noise update
// (noise.py:13-17, normal deviates from Philox + Box-Muller), stored action mu + noise, env action
// clip(a,-1,1)*high.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "ttenv.h"

namespace {

constexpr int IN = 23, INP = 24;        // input features, padded to a multiple of the MFMA k (4)
constexpr int H1 = 400, H2 = 300, H2P = 320;
constexpr int NT1 = H1 / 16;            // 25 column tiles of layer 1
constexpr int NT2 = H2P / 16;           // 20 column tiles of layer 2 (300 real + 20 zero columns)
constexpr int BM = 64, WROWS = 16;      // rows per workgroup / per wave
constexpr int HS1 = 402;                // LDS row stride of the layer-1 activations
constexpr int OS = 26;                  // LDS row stride of the obs tile and of the staged fc1 weights
constexpr int KC = 16, SK = 18;         // fc2 chunk: 16 input columns, LDS row stride 18
constexpr int NCHUNK = H1 / KC;         // 25
constexpr int WBUF = 2 * H2P * SK;      // 11520 words: two fc2 chunks; fc1 (400*26 = 10400) is staged here first
constexpr int LDS_WORDS = BM * HS1 + WBUF + BM * OS;   // 38912 words = 155,648 B (one workgroup per CU)

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Weights {
    const float *w1, *b1, *g1, *be1;    // fc1 [400,23], bias, LayerNorm weight/bias
    const float *w2, *b2, *g2, *be2;    // fc2 [300,400], ...
    const float *w3, *b3;               // head [1,300], [1]
    const float *wa, *ba;               // critic only: action_value [300,1], [300]
};

struct ActArgs {                        // optional fused choose_action epilogue (actor only)
    float *ou;                          // [n] OU state, updated in place (NULL: no noise, plain forward)
    const uint8_t *done_prev;           // [n] or NULL: envs whose episode just ended restart their noise at 0
    float *act_raw, *act_scaled;        // [n] mu + noise; clip(.,-1,1)*high
    const long long *step_dev;          // device step counter (graph-safe) or NULL
    unsigned long long seed, step;
    float decay, scale, high;           // 1 - theta*dt, sigma*sqrt(dt), action_space.high
};

__device__ inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                  uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sum over the 16 lanes that share (lane >> 4): the columns of one accumulator row
__device__ __forceinline__ float row_sum16(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

__device__ __forceinline__ void stage_w2_chunk_load(const float *w2, int c, int tid, float4 (&buf)[5]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = tid + 256 * i, j = idx >> 2, q = idx & 3;       // row j of fc2, 16-byte piece q of the chunk
        buf[i] = j < H2 ? *reinterpret_cast<const float4 *>(w2 + (size_t)j * H1 + c * KC + 4 * q)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void stage_w2_chunk_store(float *dst, int tid, const float4 (&buf)[5]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = tid + 256 * i, j = idx >> 2, q = idx & 3;
        float2 *p = reinterpret_cast<float2 *>(dst + j * SK + 4 * q);    // (72 j + 16 q) bytes: 8-byte aligned
        p[0] = make_float2(buf[i].x, buf[i].y);
        p[1] = make_float2(buf[i].z, buf[i].w);
    }
}

template <bool CRITIC>
__global__ __launch_bounds__(256, 1) void k_mlp_forward(const int n, const float *__restrict__ obs,
                                                        const float *__restrict__ action, const Weights W,
                                                        float *__restrict__ out, const ActArgs act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *h1_s = lds;                       // [64][402]
    float *wbuf = lds + BM * HS1;            // fc1 staging, then the fc2 chunk ring
    float *obs_s = wbuf + WBUF;              // [64][26]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int row0 = blockIdx.x * BM;

    // ---- stage the obs tile (zero-padded to 24 columns) and fc1's weights [400][24]
    for (int idx = tid; idx < BM * INP; idx += 256) {
        const int r = idx / INP, k = idx - r * INP;
        obs_s[r * OS + k] = (k < IN && row0 + r < n) ? obs[(size_t)(row0 + r) * IN + k] : 0.f;
    }
    for (int idx = tid; idx < H1 * INP; idx += 256) {
        const int j = idx / INP, k = idx - j * INP;
        wbuf[j * OS + k] = k < IN ? W.w1[j * IN + k] : 0.f;
    }
    __syncthreads();

    // ---- layer 1 on the MFMA: this wave's 16 rows x 400 columns, K = 24
    f32x4 acc1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < INP / 4; ++ks) {
        const float a = obs_s[(wave * WROWS + l15) * OS + ks * 4 + l4];
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const float b = wbuf[(t * 16 + l15) * OS + ks * 4 + l4];
            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1[t], 0, 0, 0);
        }
    }
    // accumulator element [t][r] is row (wave*16 + l4*4 + r), column (t*16 + l15)
    {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const float bias = W.b1[t * 16 + l15];
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc1[t][r] += bias; s[r] += acc1[t][r]; }
        }
        float mean[4], rstd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mean[r] = row_sum16(s[r]) * (1.f / H1);
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc1[t][r] - mean[r]; ss[r] = fmaf(d, d, ss[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(row_sum16(ss[r]) * (1.f / H1) + 1e-5f);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int col = t * 16 + l15;
            const float g = W.g1[col], be = W.be1[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = fmaf((acc1[t][r] - mean[r]) * rstd[r], g, be);
                h1_s[(wave * WROWS + l4 * 4 + r) * HS1 + col] = fmaxf(y, 0.f);
            }
        }
    }
    __syncthreads();   // every wave is done with the staged fc1 weights; h1 rows are wave-private

    // ---- layer 2: stream fc2 through the LDS ring, 16 input columns per chunk
    f32x4 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 pre[5];
    stage_w2_chunk_load(W.w2, 0, tid, pre);
    stage_w2_chunk_store(wbuf, tid, pre);
    for (int c = 0; c < NCHUNK; ++c) {
        __syncthreads();                                  // chunk c is in wbuf[c&1]; wbuf[(c+1)&1] is free
        if (c + 1 < NCHUNK) stage_w2_chunk_load(W.w2, c + 1, tid, pre);
        const float *wb = wbuf + (c & 1) * (H2P * SK);
        const float *arow = h1_s + (wave * WROWS + l15) * HS1 + c * KC + l4;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            const float a = arow[ks * 4];
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const float b = wb[(t * 16 + l15) * SK + ks * 4 + l4];
                acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[t], 0, 0, 0);
            }
        }
        if (c + 1 < NCHUNK) stage_w2_chunk_store(wbuf + ((c + 1) & 1) * (H2P * SK), tid, pre);
    }

    // ---- epilogue in registers: bias, LayerNorm over the 300 real columns, (critic: + action_value(a)),
    // ReLU, head, (actor: tanh)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int col = t * 16 + l15;
        const bool real = col < H2;
        const float bias = real ? W.b2[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc2[t][r] = real ? acc2[t][r] + bias : 0.f; s[r] += acc2[t][r]; }
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mean[r] = row_sum16(s[r]) * (1.f / H2);
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const bool real = t * 16 + l15 < H2;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = real ? acc2[t][r] - mean[r] : 0.f; ss[r] = fmaf(d, d, ss[r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(row_sum16(ss[r]) * (1.f / H2) + 1e-5f);
    float av[4] = {0.f, 0.f, 0.f, 0.f};
    if (CRITIC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + wave * WROWS + l4 * 4 + r;
            av[r] = row < n ? action[row] : 0.f;
        }
    }
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int col = t * 16 + l15;
        if (col < H2) {
            const float g = W.g2[col], be = W.be2[col], w3 = W.w3[col];
            float wa = 0.f, ba = 0.f;
            if (CRITIC) { wa = W.wa[col]; ba = W.ba[col]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = fmaf((acc2[t][r] - mean[r]) * rstd[r], g, be);
                if (CRITIC) y += fmaf(av[r], wa, ba);
                dot[r] = fmaf(fmaxf(y, 0.f), w3, dot[r]);
            }
        }
    }
    const float b3 = W.b3[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = row_sum16(dot[r]) + b3;
        const int row = row0 + wave * WROWS + l4 * 4 + r;
        if (l15 == 0 && row < n) {
            if (CRITIC) {
                out[row] = v;
            } else {
                const float mu = tanhf(v);
                if (out) out[row] = mu;
                if (act.ou) {
                    // OU noise (noise.py:13-17): x <- x + theta*(0 - x)*dt + sigma*sqrt(dt)*N(0,1); restart at 0
                    // for an env whose episode just ended (trainv2.py:492)
                    float x = act.ou[row];
                    if (act.done_prev && act.done_prev[row]) x = 0.f;
                    const unsigned long long st = act.step + (act.step_dev ? (unsigned long long)*act.step_dev : 0ull);
                    uint32_t rnd[4];
                    philox4x32((uint32_t)row, (uint32_t)st, (uint32_t)(st >> 32), 0x0A5Eu, (uint32_t)act.seed,
                               (uint32_t)(act.seed >> 32), rnd);
                    const float u1 = ((float)(rnd[0] >> 8) + 0.5f) * (1.f / 16777216.f);
                    const float u2 = ((float)(rnd[1] >> 8) + 0.5f) * (1.f / 16777216.f);
                    const float nrm = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                    x = fmaf(x, act.decay, act.scale * nrm);
                    act.ou[row] = x;
                    const float a = mu + x;                               // stored action: unclipped (trainv2.py:525)
                    act.act_raw[row] = a;
                    act.act_scaled[row] = fminf(fmaxf(a, -1.f), 1.f) * act.high;   // trainv2.py:516
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// v2 of the forward: no LDS weight ring and no barrier inside the K loop.  fc1 and fc2 fragments are read straight
// from L2 (they are shared by every workgroup, 37 KB + 480 KB), fc2 with the permuted-k trick: in one "k16 step"
// lane l contributes k = k0 + 4*(l>>4) + ks to MFMA ks, so its four B values are one float4 of a weight row and its
// four A values one ds_read_b128 of the activation tile.  The loads of the NEXT two k16 steps (40 float4 per lane)
// are in flight while the 160 MFMAs of the current two issue.
constexpr int HS2 = 404;     // activation tile row stride: 16-byte rows; 404 mod 64 = 20 spreads 16 rows over all banks
constexpr int V2_LDS_BYTES = (BM * HS2 + H1 * IN + 3 * H1 + 6 * H2) * 4;      // 103,424 + 36,800 + 4,800 + 7,200 = 152,224 B

template <bool CRITIC>
__global__ __launch_bounds__(256, 1) void k_mlp_forward_v2(const int n, const float *__restrict__ obs,
                                                           const float *__restrict__ action, const Weights W,
                                                           float *__restrict__ out, const ActArgs act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *h1_s = lds;                       // [64][404]; each wave only touches its own 16 rows
    float *w1_s = h1_s + BM * HS2;           // fc1 weights, raw [400][23] (stride 23 is odd: fragment reads spread over banks)
    float *p1_s = w1_s + H1 * IN;            // b1 | g1 | be1        [3][400]
    float *p2_s = p1_s + 3 * H1;             // b2 | g2 | be2 | w3 (| wa | ba)   [6][300]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int row0 = blockIdx.x * BM, wrow0 = row0 + wave * WROWS;

    // ---- one cooperative, fully coalesced copy of everything small that every wave needs: a single L2 round trip
    // instead of ~25 dependent ones (the fragment loads of layer 1 would otherwise each wait on L2)
    {
        const float4 *src = reinterpret_cast<const float4 *>(W.w1);           // 9200 floats = 2300 float4
        float4 *dst = reinterpret_cast<float4 *>(w1_s);
        for (int i = tid; i < H1 * IN / 4; i += 256) dst[i] = src[i];
        for (int i = tid; i < H1; i += 256) { p1_s[i] = W.b1[i]; p1_s[H1 + i] = W.g1[i]; p1_s[2 * H1 + i] = W.be1[i]; }
        for (int i = tid; i < H2; i += 256) {
            p2_s[i] = W.b2[i]; p2_s[H2 + i] = W.g2[i]; p2_s[2 * H2 + i] = W.be2[i]; p2_s[3 * H2 + i] = W.w3[i];
            if (CRITIC) { p2_s[4 * H2 + i] = W.wa[i]; p2_s[5 * H2 + i] = W.ba[i]; }
        }
    }
    // ---- layer 1 (K = 23)
    f32x4 acc1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        float a[INP / 4];
#pragma unroll
        for (int ks = 0; ks < INP / 4; ++ks) {
            const int k = ks * 4 + l4;
            a[ks] = (k < IN && wrow0 + l15 < n) ? obs[(size_t)(wrow0 + l15) * IN + k] : 0.f;
        }
        __syncthreads();                     // staged weights visible
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            float b[INP / 4];
#pragma unroll
            for (int ks = 0; ks < INP / 4; ++ks) {
                const int k = ks * 4 + l4;
                b[ks] = k < IN ? w1_s[(t * 16 + l15) * IN + k] : 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < INP / 4; ++ks) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[ks], acc1[t], 0, 0, 0);
        }
    }
    {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const float bias = p1_s[t * 16 + l15];
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc1[t][r] += bias; s[r] += acc1[t][r]; }
        }
        float mean[4], rstd[4], ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) mean[r] = row_sum16(s[r]) * (1.f / H1);
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc1[t][r] - mean[r]; ss[r] = fmaf(d, d, ss[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(row_sum16(ss[r]) * (1.f / H1) + 1e-5f);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int col = t * 16 + l15;
            const float g = p1_s[H1 + col], be = p1_s[2 * H1 + col];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                h1_s[(wave * WROWS + l4 * 4 + r) * HS2 + col] = fmaxf(fmaf((acc1[t][r] - mean[r]) * rstd[r], g, be), 0.f);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes have landed (rows are wave-private)
    __builtin_amdgcn_wave_barrier();

    // ---- layer 2
    f32x4 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef TT_DBG_SHORT_K
    constexpr int SS = 2, NSS = 1;     // timing experiment only: 2 of the 25 k16 steps
#else
    constexpr int SS = 2, NSS = (H1 / 16 + SS - 1) / SS;          // 25 k16 steps -> 13 super-steps (the last holds 1)
#endif
    struct BFrag { float4 v[SS][NT2]; };
    const float *wbase = W.w2 + (size_t)l15 * H1 + 4 * l4;        // + t*16*H1 per tile
    auto load_b = [&](int ss) {
        BFrag f;
#pragma unroll
        for (int st = 0; st < SS; ++st) {
            const int c = ss * SS + st;
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const bool real = c < H1 / 16 && t * 16 + l15 < H2;
                f.v[st][t] = real ? *reinterpret_cast<const float4 *>(wbase + (size_t)t * 16 * H1 + 16 * c)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        return f;
    };
    const float *arow = h1_s + (wave * WROWS + l15) * HS2 + 4 * l4;
    BFrag cur = load_b(0);
    for (int ss = 0; ss < NSS; ++ss) {
        BFrag nxt = cur;
        if (ss + 1 < NSS) nxt = load_b(ss + 1);
#pragma unroll
        for (int st = 0; st < SS; ++st) {
            const int c = ss * SS + st;
            if (c < H1 / 16) {
                const float4 av = *reinterpret_cast<const float4 *>(arow + 16 * c);
#pragma unroll
                for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, cur.v[st][t].x, acc2[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, cur.v[st][t].y, acc2[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, cur.v[st][t].z, acc2[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, cur.v[st][t].w, acc2[t], 0, 0, 0);
            }
        }
        cur = nxt;
    }

    // ---- epilogue (same as v1)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int col = t * 16 + l15;
        const bool real = col < H2;
        const float bias = real ? p2_s[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc2[t][r] = real ? acc2[t][r] + bias : 0.f; s[r] += acc2[t][r]; }
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mean[r] = row_sum16(s[r]) * (1.f / H2);
    float ss2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const bool real = t * 16 + l15 < H2;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = real ? acc2[t][r] - mean[r] : 0.f; ss2[r] = fmaf(d, d, ss2[r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(row_sum16(ss2[r]) * (1.f / H2) + 1e-5f);
    float av[4] = {0.f, 0.f, 0.f, 0.f};
    if (CRITIC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wrow0 + l4 * 4 + r;
            av[r] = row < n ? action[row] : 0.f;
        }
    }
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int col = t * 16 + l15;
        if (col < H2) {
            const float g = p2_s[H2 + col], be = p2_s[2 * H2 + col], w3 = p2_s[3 * H2 + col];
            float wa = 0.f, ba = 0.f;
            if (CRITIC) { wa = p2_s[4 * H2 + col]; ba = p2_s[5 * H2 + col]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = fmaf((acc2[t][r] - mean[r]) * rstd[r], g, be);
                if (CRITIC) y += fmaf(av[r], wa, ba);
                dot[r] = fmaf(fmaxf(y, 0.f), w3, dot[r]);
            }
        }
    }
    const float b3 = W.b3[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = row_sum16(dot[r]) + b3;
        const int row = wrow0 + l4 * 4 + r;
        if (l15 == 0 && row < n) {
            if (CRITIC) {
                out[row] = v;
            } else {
                const float mu = tanhf(v);
                if (out) out[row] = mu;
                if (act.ou) {
                    float x = act.ou[row];
                    if (act.done_prev && act.done_prev[row]) x = 0.f;
                    const unsigned long long st = act.step + (act.step_dev ? (unsigned long long)*act.step_dev : 0ull);
                    uint32_t rnd[4];
                    philox4x32((uint32_t)row, (uint32_t)st, (uint32_t)(st >> 32), 0x0A5Eu, (uint32_t)act.seed,
                               (uint32_t)(act.seed >> 32), rnd);
                    const float u1 = ((float)(rnd[0] >> 8) + 0.5f) * (1.f / 16777216.f);
                    const float u2 = ((float)(rnd[1] >> 8) + 0.5f) * (1.f / 16777216.f);
                    const float nrm = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                    x = fmaf(x, act.decay, act.scale * nrm);
                    act.ou[row] = x;
                    const float a = mu + x;
                    act.act_raw[row] = a;
                    act.act_scaled[row] = fminf(fmaxf(a, -1.f), 1.f) * act.high;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// v3: as v2 but 32 rows (2 waves) per workgroup and at most 256 registers per lane, so that three workgroups fit on
// a CU (52 KB of LDS each): one wave's LayerNorm / epilogue / load latency hides behind another wave's MFMAs.
constexpr int BM3 = 32;
constexpr int V3_LDS_BYTES = BM3 * HS2 * 4;     // 51,712 B

template <bool CRITIC>
__global__ __launch_bounds__(128, 2) void k_mlp_forward_v3(const int n, const float *__restrict__ obs,
                                                           const float *__restrict__ action, const Weights W,
                                                           float *__restrict__ out, const ActArgs act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *h1_s = lds;                       // [32][404]; each wave only touches its own 16 rows
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int row0 = blockIdx.x * BM3, wrow0 = row0 + wave * WROWS;

    // ---- layer 1 (K = 23), operands from global
    f32x4 acc1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        float a[INP / 4];
#pragma unroll
        for (int ks = 0; ks < INP / 4; ++ks) {
            const int k = ks * 4 + l4;
            a[ks] = (k < IN && wrow0 + l15 < n) ? obs[(size_t)(wrow0 + l15) * IN + k] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            float b[INP / 4];
#pragma unroll
            for (int ks = 0; ks < INP / 4; ++ks) {
                const int k = ks * 4 + l4;
                b[ks] = k < IN ? W.w1[(t * 16 + l15) * IN + k] : 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < INP / 4; ++ks) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[ks], acc1[t], 0, 0, 0);
        }
    }
    {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const float bias = W.b1[t * 16 + l15];
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc1[t][r] += bias; s[r] += acc1[t][r]; }
        }
        float mean[4], rstd[4], ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) mean[r] = row_sum16(s[r]) * (1.f / H1);
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc1[t][r] - mean[r]; ss[r] = fmaf(d, d, ss[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(row_sum16(ss[r]) * (1.f / H1) + 1e-5f);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int col = t * 16 + l15;
            const float g = W.g1[col], be = W.be1[col];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                h1_s[(wave * WROWS + l4 * 4 + r) * HS2 + col] = fmaxf(fmaf((acc1[t][r] - mean[r]) * rstd[r], g, be), 0.f);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes have landed (rows are wave-private)
    __builtin_amdgcn_wave_barrier();

    // ---- layer 2
    f32x4 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // No explicit prefetch here: two to three workgroups share a CU (52 KB of LDS each), so while this wave waits
    // for its 20 weight rows the other wave of the SIMD issues MFMAs.
    const float *wbase = W.w2 + (size_t)l15 * H1 + 4 * l4;        // + t*16*H1 per tile
    const float *arow = h1_s + (wave * WROWS + l15) * HS2 + 4 * l4;
    for (int c = 0; c < H1 / 16; ++c) {
        float4 bv[NT2];
#pragma unroll
        for (int t = 0; t < NT2; ++t)
            bv[t] = t * 16 + l15 < H2 ? *reinterpret_cast<const float4 *>(wbase + (size_t)t * 16 * H1 + 16 * c)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 av = *reinterpret_cast<const float4 *>(arow + 16 * c);
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[t].x, acc2[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[t].y, acc2[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[t].z, acc2[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[t].w, acc2[t], 0, 0, 0);
    }

    // ---- epilogue (same as v1)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int col = t * 16 + l15;
        const bool real = col < H2;
        const float bias = real ? W.b2[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc2[t][r] = real ? acc2[t][r] + bias : 0.f; s[r] += acc2[t][r]; }
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mean[r] = row_sum16(s[r]) * (1.f / H2);
    float ss2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const bool real = t * 16 + l15 < H2;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = real ? acc2[t][r] - mean[r] : 0.f; ss2[r] = fmaf(d, d, ss2[r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(row_sum16(ss2[r]) * (1.f / H2) + 1e-5f);
    float av[4] = {0.f, 0.f, 0.f, 0.f};
    if (CRITIC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wrow0 + l4 * 4 + r;
            av[r] = row < n ? action[row] : 0.f;
        }
    }
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int col = t * 16 + l15;
        if (col < H2) {
            const float g = W.g2[col], be = W.be2[col], w3 = W.w3[col];
            float wa = 0.f, ba = 0.f;
            if (CRITIC) { wa = W.wa[col]; ba = W.ba[col]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = fmaf((acc2[t][r] - mean[r]) * rstd[r], g, be);
                if (CRITIC) y += fmaf(av[r], wa, ba);
                dot[r] = fmaf(fmaxf(y, 0.f), w3, dot[r]);
            }
        }
    }
    const float b3 = W.b3[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = row_sum16(dot[r]) + b3;
        const int row = wrow0 + l4 * 4 + r;
        if (l15 == 0 && row < n) {
            if (CRITIC) {
                out[row] = v;
            } else {
                const float mu = tanhf(v);
                if (out) out[row] = mu;
                if (act.ou) {
                    float x = act.ou[row];
                    if (act.done_prev && act.done_prev[row]) x = 0.f;
                    const unsigned long long st = act.step + (act.step_dev ? (unsigned long long)*act.step_dev : 0ull);
                    uint32_t rnd[4];
                    philox4x32((uint32_t)row, (uint32_t)st, (uint32_t)(st >> 32), 0x0A5Eu, (uint32_t)act.seed,
                               (uint32_t)(act.seed >> 32), rnd);
                    const float u1 = ((float)(rnd[0] >> 8) + 0.5f) * (1.f / 16777216.f);
                    const float u2 = ((float)(rnd[1] >> 8) + 0.5f) * (1.f / 16777216.f);
                    const float nrm = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                    x = fmaf(x, act.decay, act.scale * nrm);
                    act.ou[row] = x;
                    const float a = mu + x;
                    act.act_raw[row] = a;
                    act.act_scaled[row] = fminf(fmaxf(a, -1.f), 1.f) * act.high;
                }
            }
        }
    }
}

// Uniform sampling WITH replacement from the trajectory ring (replay_buffer.py:23-34 draws np.random.choice(max_mem,
// batch)): one workgroup per sampled transition gathers s, a, r, s', done into the batch buffers.  The ring's step
// counter is read from device memory so that a captured hipGraph of learn() samples fresh indices every replay
// (the Philox counter includes it).
__global__ __launch_bounds__(64) void k_ring_sample(const int batch, const int n_envs, const int slots,
                                                    const long long *__restrict__ k_dev, const float *__restrict__ obs,
                                                    const float *__restrict__ act, const float *__restrict__ rew,
                                                    const uint8_t *__restrict__ done, const unsigned long long seed,
                                                    float *__restrict__ s_out, float *__restrict__ a_out,
                                                    float *__restrict__ r_out, float *__restrict__ s2_out,
                                                    uint8_t *__restrict__ d_out, int *__restrict__ idx_out) {
    const int b = blockIdx.x;
    if (b >= batch) return;
    const long long k = *k_dev;                       // vector steps completed; transitions k-avail .. k-1 are intact
    const long long avail = k < slots - 1 ? k : slots - 1;
    uint32_t r[4];
    philox4x32((uint32_t)b, (uint32_t)k, (uint32_t)(k >> 32), 0x5A3Du, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const long long back = avail > 0 ? (long long)(((unsigned long long)r[0] * (unsigned long long)avail) >> 32) : 0;
    const int t = (int)(((k - 1 - back) % slots + slots) % slots), t1 = (t + 1) % slots;
    const int e = (int)(((unsigned long long)r[1] * (unsigned long long)n_envs) >> 32);
    const int lane = threadIdx.x;
    const float *src = obs + ((size_t)t * n_envs + e) * IN, *src2 = obs + ((size_t)t1 * n_envs + e) * IN;
    if (lane < IN) s_out[(size_t)b * IN + lane] = src[lane];
    else if (lane >= 32 && lane < 32 + IN) s2_out[(size_t)b * IN + lane - 32] = src2[lane - 32];
    if (lane == 63) {
        const size_t q = (size_t)t * n_envs + e;
        a_out[b] = act[q];
        r_out[b] = rew[q];
        d_out[b] = done[q];
        if (idx_out) { idx_out[2 * b] = t; idx_out[2 * b + 1] = e; }
    }
}

int check_ptrs(const tt_mlp_weights *w, bool critic) {
    if (!w) return 0;
    if (w->in_dim != IN || w->fc1_dims != H1 || w->fc2_dims != H2) return 0;
    if (!w->w1 || !w->b1 || !w->g1 || !w->be1 || !w->w2 || !w->b2 || !w->g2 || !w->be2 || !w->w3 || !w->b3) return 0;
    if (critic && (!w->wa || !w->ba)) return 0;
    return 1;
}

Weights to_weights(const tt_mlp_weights *w) {
    return Weights{w->w1, w->b1, w->g1, w->be1, w->w2, w->b2, w->g2, w->be2, w->w3, w->b3, w->wa, w->ba};
}

int mlp_version() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("TT_MLP_VERSION");       // A/B switch for measurements; 2 = direct-from-L2 weights
        v = (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 2;
    }
    return v;
}

template <bool CRITIC>
int launch(int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out, const ActArgs &act,
           hipStream_t stream) {
    if (mlp_version() == 3) {
        static bool attr3 = false;
        if (!attr3) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_forward_v3<CRITIC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS_BYTES) != hipSuccess)
                return TT_EHIP;
            attr3 = true;
        }
        hipLaunchKernelGGL((k_mlp_forward_v3<CRITIC>), dim3((n + BM3 - 1) / BM3), dim3(128), V3_LDS_BYTES, stream, n, obs,
                           action, to_weights(w), out, act);
        return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
    }
    if (mlp_version() == 2) {
        static bool attr2 = false;
        if (!attr2) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_forward_v2<CRITIC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES) != hipSuccess)
                return TT_EHIP;
            attr2 = true;
        }
        hipLaunchKernelGGL((k_mlp_forward_v2<CRITIC>), dim3((n + BM - 1) / BM), dim3(256), V2_LDS_BYTES, stream, n, obs,
                           action, to_weights(w), out, act);
        return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_forward<CRITIC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WORDS * 4) != hipSuccess)
            return TT_EHIP;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_mlp_forward<CRITIC>), dim3((n + BM - 1) / BM), dim3(256), LDS_WORDS * 4, stream, n, obs, action,
                       to_weights(w), out, act);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

}  // namespace

extern "C" {

int tt_actor_forward(int n, const float *obs, const tt_mlp_weights *w, float *mu_out, tt_stream_t stream) {
    if (n < 0 || !obs || !mu_out || !check_ptrs(w, false)) return TT_EINVAL;
    if (n == 0) return TT_OK;
    ActArgs act{};
    return launch<false>(n, obs, nullptr, w, mu_out, act, stream);
}

int tt_actor_act(int n, const float *obs, const tt_mlp_weights *w, float *ou_state, const uint8_t *done_prev,
                 uint64_t seed, uint64_t step, const int64_t *step_dev, float theta_dt, float sigma_sqrt_dt, float high,
                 float *mu_out, float *act_raw_out, float *act_scaled_out, tt_stream_t stream) {
    if (n < 0 || !obs || !ou_state || !act_raw_out || !act_scaled_out || !check_ptrs(w, false)) return TT_EINVAL;
    if (n == 0) return TT_OK;
    ActArgs act{};
    act.ou = ou_state; act.done_prev = done_prev; act.act_raw = act_raw_out; act.act_scaled = act_scaled_out;
    act.step_dev = reinterpret_cast<const long long *>(step_dev);
    act.seed = seed; act.step = step;
    act.decay = 1.0f - theta_dt; act.scale = sigma_sqrt_dt; act.high = high;
    return launch<false>(n, obs, nullptr, w, mu_out, act, stream);
}

int tt_ring_sample(int batch, int n_envs, int slots, const int64_t *k_dev, const float *obs, const float *act,
                   const float *rew, const uint8_t *done, uint64_t seed, float *s_out, float *a_out, float *r_out,
                   float *s2_out, uint8_t *d_out, int32_t *idx_out, tt_stream_t stream) {
    if (batch < 0 || n_envs <= 0 || slots < 3 || !k_dev || !obs || !act || !rew || !done || !s_out || !a_out || !r_out ||
        !s2_out || !d_out)
        return TT_EINVAL;
    if (batch == 0) return TT_OK;
    hipLaunchKernelGGL(k_ring_sample, dim3(batch), dim3(64), 0, stream, batch, n_envs, slots,
                       reinterpret_cast<const long long *>(k_dev), obs, act, rew, done, seed, s_out, a_out, r_out, s2_out,
                       d_out, idx_out);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_critic_forward(int n, const float *obs, const float *action, const tt_mlp_weights *w, float *q_out,
                      tt_stream_t stream) {
    if (n < 0 || !obs || !action || !q_out || !check_ptrs(w, true)) return TT_EINVAL;
    if (n == 0) return TT_OK;
    ActArgs act{};
    return launch<true>(n, obs, action, w, q_out, act, stream);
}

}  // extern "C"
