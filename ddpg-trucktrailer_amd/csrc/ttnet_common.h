// ttnet_common.h -- what the inference kernels of csrc/ttnet.hip (exact-f32 MFMA) and csrc/ttnet_split.hip (split-bf16
// MFMA) share: the network shapes of trainv2.py:404-407, the kernel-side weight / choose_action argument structs, the
// Philox generator and the per-row tail of the forward (head output -> tanh -> OU noise -> clip*high).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ttenv.h"

namespace ttnet {

constexpr int IN = 23;                  // observation features (simv2.py:79-84)
constexpr int H1 = 400, H2 = 300;       // fc1_dims, fc2_dims

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

// The tail of one row once the head's pre-activation v is known.  Critic: out[row] = v.  Actor: mu = tanh(v)
// (networks.py:145) and, when act.ou is set, DDPG_agent.choose_action + trainv2.py:516: OU noise (noise.py:13-17:
// x <- x + theta*(0 - x)*dt + sigma*sqrt(dt)*N(0,1), restarted at 0 for an env whose episode just ended,
// trainv2.py:492; N(0,1) from Philox + Box-Muller), stored action mu + x (unclipped, trainv2.py:525), env action
// clip(a,-1,1)*high.
template <bool CRITIC>
__device__ __forceinline__ void finish_row(const int row, const float v, float *__restrict__ out, const ActArgs &act) {
    if (CRITIC) {
        out[row] = v;
        return;
    }
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

inline Weights to_weights(const tt_mlp_weights *w) {
    return Weights{w->w1, w->b1, w->g1, w->be1, w->w2, w->b2, w->g2, w->be2, w->w3, w->b3, w->wa, w->ba};
}

// csrc/ttnet_split.hip
size_t split_ws_bytes();
int split_pack(const tt_mlp_weights *w, bool critic, void *ws, long long *bump, hipStream_t stream);
int split_forward(bool critic, int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                  const ActArgs &act, hipStream_t stream);
#ifdef TT_STAMPS
int split_debug_stamps(unsigned long long *out16);
int split_debug_block_stamps(unsigned long long *out, int nblocks);
#endif

}  // namespace ttnet
