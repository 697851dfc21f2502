// ttnet.hip -- fused inference of the reference's actor / critic MLPs on MI355X (gfx950), exact f32.
//
//   actor  (DDPG/networks.py:138-147): tanh(mu(relu(bn2(fc2(relu(bn1(fc1(s))))))))
//   critic (DDPG/networks.py:55-68):   q(relu(bn2(fc2(relu(bn1(fc1(s))))) + action_value(a)))
// with fc1 23->400, fc2 400->300, LayerNorm (eps 1e-5, biased variance) -- the shapes of trainv2.py:404-407.
//
// k_mlp_forward_v2 (this file) is the exact-f32 kernel: one workgroup (4 waves) pushes a tile of 64 rows through the
// whole network on the f32-input MFMA (v_mfma_f32_16x16x4_f32: a k-ordered fma chain); each wave owns 16 rows and
// ALL output columns, so both LayerNorms and the final dot product reduce inside the wave.  It serves small batches
// and is the bit-reference of the split-bf16 kernel of csrc/ttnet_split.hip, which the launcher prefers for large
// batches when the caller provides its workspace (tt_mlp_weights.split_ws).
//
// The actor kernels can also finish DDPG_agent.choose_action + trainv2.py:516 in their epilogue (ttnet_common.h:
// finish_row).  k_ring_sample (ReplayBuffer.sample_buffer on the device trajectory ring) lives here as well.
#include <cstdio>
#include <cstdlib>

#include "ttnet_common.h"

using namespace ttnet;

namespace {

constexpr int INP = 24;                 // input features padded to a multiple of the MFMA k (4)
constexpr int H2P = 320;
constexpr int NT1 = H1 / 16;            // 25 column tiles of layer 1
constexpr int NT2 = H2P / 16;           // 20 column tiles of layer 2 (300 real + 20 zero columns)
constexpr int BM = 64, WROWS = 16;      // rows per workgroup / per wave

using f32x4 = __attribute__((ext_vector_type(4))) float;

// sum over the 16 lanes that share (lane >> 4): the columns of one accumulator row.  On the DPP path (VALU speed);
// __shfl_xor would be a ds_bpermute, an LDS round trip per step
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f<0xB1>(v);       // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_f<0x4E>(v);       // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_f<0x141>(v);      // row_half_mirror: the other quad of the half row
    v += dpp_f<0x140>(v);      // row_mirror: the other half row
    return v;
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
            a[ks] = (k < IN && wrow0 + l15 < n) ? resolve_obs(act, obs)[(size_t)(wrow0 + l15) * IN + k] : 0.f;
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

    // ---- epilogue in registers: bias, LayerNorm over the 300 real columns, (critic: + action_value(a)), ReLU, head
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
        if (l15 == 0 && row < n) finish_row<CRITIC>(row, v, out, act);
    }
}

int check_ptrs(const tt_mlp_weights *w, bool critic) {
    if (!w) return 0;
    if (w->in_dim != IN || w->fc1_dims != H1 || w->fc2_dims != H2) return 0;
    if (!w->w1 || !w->b1 || !w->g1 || !w->be1 || !w->w2 || !w->b2 || !w->g2 || !w->be2 || !w->w3 || !w->b3) return 0;
    if (critic && (!w->wa || !w->ba)) return 0;
    return 1;
}


// which kernel serves a forward of n rows: the split-f16 kernel needs the caller's workspace and pays a weight-pack
// launch per call, so it takes over from 1024 rows; TT_MLP_KERNEL=f32 / split forces one (A/B measurements, tests)
bool use_split(int n, const tt_mlp_weights *w) {
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("TT_MLP_KERNEL");
        forced = !e ? 0 : (e[0] == 'f' ? 1 : (e[0] == 's' ? 2 : 0));
    }
    if (!w->split_ws) return false;
    if (w->ws_packed) return true;      // the caller keeps an image BECAUSE the live weights may be changing: never read them
    if (forced == 1) return false;
    return forced == 2 || n >= 1024;
}

template <bool CRITIC>
int launch(int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out, const ActArgs &act,
           hipStream_t stream) {
    if (use_split(n, w)) {
        // ws_packed: the caller keeps the image current itself (tt_mlp_split_pack after every weight update)
        const int rc = w->ws_packed ? TT_OK : split_pack(w, CRITIC, w->split_ws, nullptr, RingCursor{nullptr, 0, nullptr}, stream);
        return rc != TT_OK ? rc : split_forward(CRITIC, n, obs, action, w, out, act, stream);
    }
    static bool attr2[64] = {};     // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TT_EHIP;
    if (!attr2[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_forward_v2<CRITIC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES) != hipSuccess)
            return TT_EHIP;
        attr2[dev] = true;
    }
    hipLaunchKernelGGL((k_mlp_forward_v2<CRITIC>), dim3((n + BM - 1) / BM), dim3(256), V2_LDS_BYTES, stream, n, obs,
                       action, to_weights(w), out, act);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

__global__ __launch_bounds__(64) void k_ring_sample(const RingSample R) {
    if ((int)blockIdx.x < R.batch) ring_sample_row(R, blockIdx.x, threadIdx.x);
}

}  // namespace

extern "C" {

#ifdef TT_STAMPS
int tt_debug_nstamps(unsigned long long *out16) { return ttnet::split_debug_stamps(out16); }
int tt_debug_bstamps(unsigned long long *out, int nblocks) { return ttnet::split_debug_block_stamps(out, nblocks); }
int tt_debug_policy_tiles(unsigned long long *out4096) { return ttnet::split_debug_tiles(out4096); }
int tt_debug_policy_poll(unsigned long long *out4) { return ttnet::split_debug_poll(out4); }
int tt_debug_log_policy(unsigned long long *out, int reset) { return ttnet::split_debug_log(out, reset); }
#endif

uint64_t tt_mlp_split_ws_bytes(void) { return (uint64_t)split_ws_bytes(); }

static RingCursor to_cursor(const tt_ring_cursor *c) {
    return (c && c->cursor && c->k_dev && c->slots > 0)
               ? RingCursor{reinterpret_cast<const long long *>(c->k_dev), c->slots, c->cursor} : RingCursor{nullptr, 0, nullptr};
}

int tt_mlp_split_pack(const tt_mlp_weights *w, int critic, void *ws, int64_t *bump, const tt_ring_cursor *cursor,
                      tt_stream_t stream) {
    if (!ws || !check_ptrs(w, critic != 0)) return TT_EINVAL;
    if (bump && cursor && bump == cursor->k_dev) return TT_EINVAL;      // (the launch reads the cursor's counter to its end)
    return split_pack(w, critic != 0, ws, reinterpret_cast<long long *>(bump), to_cursor(cursor), stream);
}

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

static int make_sample(int batch, int n_envs, int slots, const int64_t *k_dev, const float *obs, const float *act,
                       const float *rew, const uint8_t *done, uint64_t seed, int reserve, int lag, const tt_side_buffer *side,
                       float *s_out, float *a_out, float *r_out, float *s2_out, uint8_t *d_out, int32_t *idx_out,
                       RingSample &R) {
    const tt_sample_args a{batch, n_envs, slots, reserve, k_dev, obs, act, rew, done, seed, side, s_out, a_out, r_out, s2_out,
                           d_out, idx_out, lag, 1, 0, nullptr};
    return make_ring_sample(&a, R);
}

int tt_actor_act_ring(int n, const tt_ring_view *ring, const tt_mlp_weights *w, float *ou_state, uint64_t seed, uint64_t step,
                      const int64_t *step_dev, float theta_dt, float sigma_sqrt_dt, float high, float *act_scaled_out,
                      tt_stream_t stream) {
    if (n <= 0 || !ring || !ring->cursor || !ring->obs || !ring->act || !ring->done || ring->n_envs != n || !ou_state ||
        !step_dev || !act_scaled_out || !check_ptrs(w, false) || !w->split_ws || !w->ws_packed)
        return TT_EINVAL;          // (ring addressing goes with a caller-kept image: the pack launch writes the cursor)
    ActArgs act{};
    act.ou = ou_state; act.done_prev = ring->done; act.act_raw = ring->act; act.act_scaled = act_scaled_out;
    act.step_dev = reinterpret_cast<const long long *>(step_dev);
    act.seed = seed; act.step = step;
    act.decay = 1.0f - theta_dt; act.scale = sigma_sqrt_dt; act.high = high;
    act.cursor = ring->cursor; act.ring_n = n; act.ring_slots = ring->slots;
    return launch<false>(n, ring->obs, nullptr, w, nullptr, act, stream);
}

int tt_ring_sample(int batch, int n_envs, int slots, const int64_t *k_dev, const float *obs, const float *act,
                   const float *rew, const uint8_t *done, uint64_t seed, int reserve, int lag, const tt_side_buffer *side,
                   float *s_out, float *a_out, float *r_out, float *s2_out, uint8_t *d_out, int32_t *idx_out,
                   tt_stream_t stream) {
    RingSample R;
    const int rc = make_sample(batch, n_envs, slots, k_dev, obs, act, rew, done, seed, reserve, lag, side, s_out, a_out, r_out,
                               s2_out, d_out, idx_out, R);
    if (rc != TT_OK || batch == 0) return rc;
    hipLaunchKernelGGL(k_ring_sample, dim3(batch), dim3(64), 0, stream, R);
    return hipGetLastError() == hipSuccess ? TT_OK : TT_EHIP;
}

int tt_mlp_split_pack_and_sample(const tt_mlp_weights *w, int critic, void *ws, const tt_sample_args *a,
                                 const tt_ring_cursor *cursor, tt_stream_t stream) {
    if (!ws || !a || !check_ptrs(w, critic != 0)) return TT_EINVAL;
    RingSample R;
    const int rc = make_sample(a->batch, a->n_envs, a->slots, a->k_dev, a->obs, a->act, a->rew, a->done, a->seed, a->reserve,
                               a->lag, a->side, a->s_out, a->a_out, a->r_out, a->s2_out, a->d_out, a->idx_out, R);
    if (rc != TT_OK) return rc;
    if (a->draws < 0 || (long long)(a->draws > 1 ? a->draws : 1) * a->batch > (1 << 24)) return TT_EINVAL;
    if (a->draws > 1) { R.draws = a->draws; R.seed_stride = a->seed_stride; }
    return split_pack_and_sample(w, critic != 0, ws, R, to_cursor(cursor), stream);
}

int tt_critic_forward(int n, const float *obs, const float *action, const tt_mlp_weights *w, float *q_out,
                      tt_stream_t stream) {
    if (n < 0 || !obs || !action || !q_out || !check_ptrs(w, true)) return TT_EINVAL;
    if (n == 0) return TT_OK;
    ActArgs act{};
    return launch<true>(n, obs, action, w, q_out, act, stream);
}

}  // extern "C"

