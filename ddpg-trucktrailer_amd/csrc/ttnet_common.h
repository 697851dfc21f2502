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
    // Optional ring addressing (tt_actor_act_ring): obs / act_raw / done_prev are then the BASES of the trajectory ring's
    // obs [slots,n,23], act [slots,n], done [slots,n] and the slot comes from the device cursor {t, t+1, t-1, t > 0} that the
    // step's opening pack launch wrote -- so one captured launch serves every ring position.  The cursor buffer holds
    // three of them: [4..7] written for even steps, [8..11] for odd steps (the opening launch of step t+1 may run beside the
    // launches of step t), [0..3] the running step's copy that this launch leaves for the env step.  step_dev is then
    // the ring's step counter: its parity picks the pair, and (tt_mlp_weights.split_ws_alt) the policy image.
    int *cursor;
    int ring_n;
    int ring_slots;                     // slots of the ring (tt_ring_view): the running step's slot numbers follow from *step_dev
};

// the running step's cursor: the one of the parity of the ring's step counter
__device__ __forceinline__ const int *cursor_of(const ActArgs &act) {
    return act.cursor + 4 + 4 * (int)(*act.step_dev & 1);
}
// obs rows of this forward: the pointer itself, or slot cursor[0] of the ring
__device__ __forceinline__ const float *resolve_obs(const ActArgs &act, const float *obs) {
    return act.cursor ? obs + (size_t)cursor_of(act)[0] * act.ring_n * IN : obs;
}

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
        const uint8_t *done_prev = act.done_prev;
        float *act_raw = act.act_raw;
        if (act.cursor) {
            const int *c = cursor_of(act);
            done_prev = c[3] ? act.done_prev + (size_t)c[2] * act.ring_n : nullptr;
            act_raw += (size_t)c[0] * act.ring_n;
        }
        if (done_prev && done_prev[row]) x = 0.f;
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
        act_raw[row] = a;
        act.act_scaled[row] = fminf(fmaxf(a, -1.f), 1.f) * act.high;
    }
}

// finish_row in two pieces, for a kernel that can ask for the row's noise inputs long before the head's output exists (the split
// kernel, per tile): everything of the OU update -- the state, the done flag of the previous step, Philox, Box-Muller -- is
// independent of the network.  ou_advance() is noise.py:13-17 + trajectory restart (trainv2.py:492) on values loaded earlier;
// finish_row_noise() is what is left once v is known: tanh, the sum, three stores.  Same arithmetic, same bits as finish_row.
__device__ __forceinline__ float ou_advance(const int row, const float x_prev, const bool restart, const ActArgs &act,
                                            const unsigned long long st) {
    float x = restart ? 0.f : x_prev;
    uint32_t rnd[4];
    philox4x32((uint32_t)row, (uint32_t)st, (uint32_t)(st >> 32), 0x0A5Eu, (uint32_t)act.seed, (uint32_t)(act.seed >> 32), rnd);
    const float u1 = ((float)(rnd[0] >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(rnd[1] >> 8) + 0.5f) * (1.f / 16777216.f);
    const float nrm = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    return fmaf(x, act.decay, act.scale * nrm);
}
__device__ __forceinline__ void finish_row_noise(const int row, const float v, float *__restrict__ out, const ActArgs &act,
                                                 const float x_new, float *__restrict__ act_raw) {
    const float mu = tanhf(v);
    if (out) out[row] = mu;
    act.ou[row] = x_new;
    const float a = mu + x_new;
    act_raw[row] = a;
    act.act_scaled[row] = fminf(fmaxf(a, -1.f), 1.f) * act.high;
}


// ------------------------------------------------------------------------------------------------------
// Uniform sampling WITH replacement from the trajectory ring (replay_buffer.py:23-34 draws np.random.choice(max_mem,
// batch)): 64 lanes gather one sampled transition s, a, r, s', done into the batch buffers.  The ring's step counter is
// read from device memory so that a captured hipGraph of learn() samples fresh indices every replay (the Philox counter
// includes it).  An optional side buffer of stand-alone transitions (expert tuples re-inserted the way trainv2.py:457-466
// `remember`s them) is part of the same uniform draw: with M side transitions and R intact ring transitions each of the
// M + R is picked with probability 1/(M + R).  Used by k_ring_sample (csrc/ttnet.hip) and by the pack-and-sample launch
// that opens a pipelined vector step (csrc/ttnet_split.hip).
struct SideBuf {
    const float *obs, *act, *rew, *obs2;
    const uint8_t *done;
    int count;
};
struct RingSample {
    int batch, n_envs, slots, reserve, lag;
    const long long *k_dev;
    const float *obs, *act, *rew;
    const uint8_t *done;
    unsigned long long seed;
    SideBuf side;
    float *s_out, *a_out, *r_out, *s2_out;
    uint8_t *d_out;
    int *idx_out;
    int draws;                         // draws of `batch` rows each in one launch (k_pack_and_sample only; else 1)
    unsigned long long seed_stride;    // draw u uses seed + u * seed_stride
    int *progress;                     // (k_fwd_multi, sampled) nullptr or the step chain's progress word to wait for (await_progress)
};

// where batch row b of a draw comes from: a side tuple j (side = true) or ring transition (slot t, env e), t1 = the slot of s'
struct RingPick {
    bool side;
    int j, t, t1, e;
};
__device__ inline RingPick ring_sample_index(const RingSample &R, const int b, const unsigned long long seed) {
    const int n_envs = R.n_envs, slots = R.slots;
    // vector steps completed; transitions k-avail .. k-1 are intact.  lag: that many of the newest steps may still be under
    // way on another stream when this draw runs (a loop whose learn() chain runs ahead of its env steps)
    const long long k0 = *R.k_dev - R.lag, k = k0 > 0 ? k0 : 0;
    const long long cap = slots - 1 - R.reserve;      // reserve: slots a concurrent env step is overwriting (pipelined loop)
    const long long avail = k < cap ? k : cap;
    uint32_t r[4];
    philox4x32((uint32_t)b, (uint32_t)k, (uint32_t)(k >> 32), 0x5A3Du, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    RingPick p{false, 0, 0, 0, 0};
    if (R.side.count > 0) {
        const unsigned long long in_ring = (unsigned long long)avail * (unsigned long long)n_envs;
        const unsigned long long u = ((unsigned long long)r[2] << 32) | r[3];
        if (__umul64hi(u, in_ring + (unsigned long long)R.side.count) < (unsigned long long)R.side.count) {
            p.side = true;
            p.j = (int)(((unsigned long long)r[0] * (unsigned long long)R.side.count) >> 32);
            return p;
        }
    }
    const long long back = avail > 0 ? (long long)(((unsigned long long)r[0] * (unsigned long long)avail) >> 32) : 0;
    p.t = (int)(((k - 1 - back) % slots + slots) % slots);
    p.t1 = (p.t + 1) % slots;
    p.e = (int)(((unsigned long long)r[1] * (unsigned long long)n_envs) >> 32);
    return p;
}
__device__ inline RingPick ring_sample_index(const RingSample &R, const int b) { return ring_sample_index(R, b, R.seed); }
// the picked transition's pieces
__device__ __forceinline__ const float *ring_pick_s(const RingSample &R, const RingPick &p) {
    return p.side ? R.side.obs + (size_t)p.j * IN : R.obs + ((size_t)p.t * R.n_envs + p.e) * IN;
}
__device__ __forceinline__ const float *ring_pick_s2(const RingSample &R, const RingPick &p) {
    return p.side ? R.side.obs2 + (size_t)p.j * IN : R.obs + ((size_t)p.t1 * R.n_envs + p.e) * IN;
}
__device__ __forceinline__ float ring_pick_a(const RingSample &R, const RingPick &p) {
    return p.side ? R.side.act[p.j] : R.act[(size_t)p.t * R.n_envs + p.e];
}
__device__ __forceinline__ float ring_pick_r(const RingSample &R, const RingPick &p) {
    return p.side ? R.side.rew[p.j] : R.rew[(size_t)p.t * R.n_envs + p.e];
}
__device__ __forceinline__ uint8_t ring_pick_d(const RingSample &R, const RingPick &p) {
    return p.side ? R.side.done[p.j] : R.done[(size_t)p.t * R.n_envs + p.e];
}

// b: row of the output buffers = draw (b / batch), row (b % batch) of that draw
__device__ inline void ring_sample_row(const RingSample &R, const int b, const int lane) {
    const int u = R.draws > 1 ? b / R.batch : 0;
    const RingPick p = ring_sample_index(R, b - u * R.batch, R.seed + (unsigned long long)u * R.seed_stride);
    const float *src = ring_pick_s(R, p), *src2 = ring_pick_s2(R, p);
    if (lane < IN) R.s_out[(size_t)b * IN + lane] = src[lane];
    else if (lane >= 32 && lane < 32 + IN) R.s2_out[(size_t)b * IN + lane - 32] = src2[lane - 32];
    if (lane == 63) {
        R.a_out[b] = ring_pick_a(R, p);
        R.r_out[b] = ring_pick_r(R, p);
        R.d_out[b] = ring_pick_d(R, p);
        if (R.idx_out) { R.idx_out[2 * b] = p.side ? -1 : p.t; R.idx_out[2 * b + 1] = p.side ? p.j : p.e; }
    }
}

// tt_sample_args -> the kernel-side struct (TT_EINVAL on a missing buffer or an impossible window)
inline int make_ring_sample(const tt_sample_args *a, RingSample &R) {
    if (!a || a->batch < 0 || a->n_envs <= 0 || a->reserve < 0 || a->lag < 0 || a->slots < 3 + a->reserve || !a->k_dev || !a->obs ||
        !a->act || !a->rew || !a->done || !a->s_out || !a->a_out || !a->r_out || !a->s2_out || !a->d_out)
        return TT_EINVAL;
    SideBuf sb{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (a->side && a->side->count > 0) {
        const tt_side_buffer *sd = a->side;
        if (!sd->obs || !sd->act || !sd->rew || !sd->obs2 || !sd->done) return TT_EINVAL;
        sb = SideBuf{sd->obs, sd->act, sd->rew, sd->obs2, sd->done, sd->count};
    }
    R = RingSample{a->batch, a->n_envs, a->slots, a->reserve, a->lag, reinterpret_cast<const long long *>(a->k_dev), a->obs, a->act,
                   a->rew, a->done, a->seed, sb, a->s_out, a->a_out, a->r_out, a->s2_out, a->d_out, a->idx_out, 1, 0ull, nullptr};
    return TT_OK;
}

inline Weights to_weights(const tt_mlp_weights *w) {
    return Weights{w->w1, w->b1, w->g1, w->be1, w->w2, w->b2, w->g2, w->be2, w->w3, w->b3, w->wa, w->ba};
}

// what the opening pack launch of a vector step writes for the step's other launches: {t, t+1, t-1, t > 0} (mod slots)
struct RingCursor {
    const long long *k_dev;
    int slots;
    int *cursor;
};
__device__ __forceinline__ void write_cursor(const RingCursor &c) {
    if (!c.cursor) return;
    const long long k = *c.k_dev;
    int *out = c.cursor + 4 + 4 * (int)(k & 1);      // the pair of this step's parity (ActArgs above)
    out[0] = (int)(k % c.slots);
    out[1] = (int)((k + 1) % c.slots);
    out[2] = (int)((k + c.slots - 1) % c.slots);
    out[3] = k > 0 ? 1 : 0;
}

// ---- the hand-over "image + cursor of step k are complete" from the opening pack launch to the policy launch of step k,
// through device memory instead of a dependency between the two launches.  Words of the cursor buffer (16 ints):
//   [12], [13]  epoch of the image / cursor pair of even / odd steps: k + 1 once the pack launch of step k has finished;
//   [14]        workgroups of the running pack launch that have finished (back to 0 by the last one);
//   [15]        set (to k + 1) by a policy launch that gave up waiting (TT_IMAGE_WAIT_TICKS of the 100 MHz clock).
// Why: in a loop of two chains of launches (policy + env step | learn()), the one launch-to-launch dependency per step from
// the learn chain's pack into the step chain's policy cost 9 us of every 92 us vector step on MI355X although it was always
// satisfied long before (the queue still stops at the wait packet).  Protocol (MI355X_MICROARCH.md, inter-workgroup
// visibility): producer -- every storing wave waits for its stores, workgroup barrier, ONE lane: agent-scope release, wait,
// agent-scope add on [14]; the workgroup whose add returns (workgroups - 1) stores the epoch.  Consumer -- one lane polls the
// epoch (relaxed, agent scope, s_sleep between polls, bounded), then agent-scope acquire, wait, workgroup barrier; only then
// does the workgroup read cursor or image.  Both sides run on every ring-addressed launch; where the two launches are ordered
// anyway (same stream) the first poll succeeds.
#ifdef TT_STAMPS      // diagnostic build: how the two hand-over waits ended, per workgroup -- [0] image: first poll succeeded, [1] image:
// had to wait, [2] progress: first poll, [3] progress: waited (one copy per translation unit: the policy's in ttnet_split.hip,
// learn()'s in ttlearn.hip)
static __device__ unsigned long long g_poll[4];
#define TT_POLL(i) atomicAdd(&g_poll[i], 1ull)
#else
#define TT_POLL(i) do { } while (0)
#endif
constexpr int CUR_EPOCH = 12, CUR_ARRIVED = 14, CUR_GAVE_UP = 15, CUR_PROGRESS = 16, CUR_GAVE_UP_MIRROR = 18;
constexpr unsigned long long TT_IMAGE_WAIT_TICKS = 25000000ull;      // 0.25 s
// (Tried: write-through sc1 stores of image and cursor + each wave's wait, no release fence -- cheaper for the pack launch, but
// tests/test_distributed.py::test_two_rank_loop_graphs_match_eager then saw a stale image on a plain kernel-to-kernel
// boundary: kept to plain stores + ONE agent-scope release per workgroup.)
// A launch that gives up marks cursor[CUR_GAVE_UP] and, when the caller left the address of a word of device-visible HOST memory in
// cursor[CUR_GAVE_UP_MIRROR .. + 1] (64 bits, 0 = none), that word as well (system scope): the host can then see the mark by reading
// its own memory -- no copy, no synchronize in the loop (DDPGRollout.run checks it after every graph replay).
__device__ __forceinline__ void mark_gave_up(int *cursor, const int value) {
    __hip_atomic_store(cursor + CUR_GAVE_UP, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int *mirror = *reinterpret_cast<int *const *>(cursor + CUR_GAVE_UP_MIRROR);
    if (mirror) __hip_atomic_store(mirror, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void publish_image(const RingCursor &c, const int workgroups) {      // every thread of the block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && c.cursor) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler may drop its own wait behind the release)
        const int before = __hip_atomic_fetch_add(c.cursor + CUR_ARRIVED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == workgroups - 1) {
            const long long k = *c.k_dev;
            __hip_atomic_store(c.cursor + CUR_ARRIVED, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.cursor + CUR_EPOCH + (int)(k & 1), (int)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// early: both epoch words as thread 0 found them BEFORE it knew the step number (await_image_early: loads that do not depend on
// the step counter, so that they share its round trip instead of following it); {0, 0} from the other threads / without a cursor
struct EpochPair { int e[2]; };
__device__ __forceinline__ EpochPair await_image_early(int *cursor) {
    EpochPair p{{0, 0}};
    if (cursor && threadIdx.x == 0) {
        p.e[0] = __hip_atomic_load(cursor + CUR_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.e[1] = __hip_atomic_load(cursor + CUR_EPOCH + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return p;
}
__device__ __forceinline__ void await_image(int *cursor, const long long *step_dev, const EpochPair early = EpochPair{{0, 0}}) {             // every thread of the block
    if (!cursor) return;
    if (threadIdx.x == 0) {
        const long long k = *step_dev;
        const int want = (int)(k + 1);
        const int *flag = cursor + CUR_EPOCH + (int)(k & 1);
        // Fast path (what a loop in step sees): the epoch is already there and no acquire is executed (a fence here costs ~1.7 us on
        // the critical path of every policy launch).  INVARIANT this relies on: between the BEGIN of this dispatch -- whose
        // acquire invalidated every L1 and the non-local lines of every XCD's L2 -- and the publish of the epoch, NOTHING on the
        // device reads the image / cursor pair of this parity: this launch's own workgroups read it only after they have seen the
        // epoch (here), a workgroup that had to wait acquires below before it reads, and the only other reader of this parity's
        // image is the policy launch of step k - 2, which is over before the pack of step k starts (the step chain's progress
        // word orders that).  So also a workgroup that starts late and finds an epoch that was published in the middle of this
        // dispatch meets no line of the image that was cached before the publish.  The pack launch of step k + 2 is the next
        // writer.  Checked by the 1500-step bitwise test and the soaks (tools/soak.py), not provable from here.
        // (`early`: the same word as read a round trip earlier, beside the step counter; epochs only grow, so an early value that
        // is large enough is as good as a fresh one)
        if (early.e[k & 1] - want >= 0 || __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want >= 0) TT_POLL(0);
        else {
            TT_POLL(1);
            // (one poll per ~3 us and workgroup: 171 workgroups polling one word every 0.1 us slowed the very launches they
            // were waiting for -- seen under rocprofv3's kernel trace, where the learn chain falls behind: 224 us per policy launch)
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
                __builtin_amdgcn_s_sleep(127);
                if (wall_clock64() - t0 > TT_IMAGE_WAIT_TICKS) {      // never hang: leave a mark the host checks, and go on
                    mark_gave_up(cursor, want);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
}

// ---- the other direction: "the step chain has reached step p" = cursor[CUR_PROGRESS] = p + 1, stored by the first workgroup
// of the policy launch of step p when it starts (everything in front of it on its stream -- the env step of step p - 1 -- is
// then over and written back).  The first launch of a pipelined learn() waits for progress >= its window counter instead of for a
// graph edge from that env step (3.4 us of every 92 us step at N = 65536: the wait packet of an edge that was always
// satisfied).  Same protocol and limits as await_image; progress points at cursor + CUR_PROGRESS.
__device__ __forceinline__ void await_progress(int *progress, const long long *k_dev) {          // every thread of the block
    if (!progress) return;
    if (threadIdx.x == 0) {
        const int want = (int)*k_dev;
        if (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want >= 0) TT_POLL(2);
        else {
            TT_POLL(3);
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
                __builtin_amdgcn_s_sleep(32);
                if (wall_clock64() - t0 > TT_IMAGE_WAIT_TICKS) {
                    mark_gave_up(progress - CUR_PROGRESS, want + 1);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    // (the acquire leaves the SCALAR cache alone -- DESIGN.md section 4.2 (i) -- but nothing the step chain writes is read through
    // it here: the ring's rows come by vector loads, the window counter is the learn chain's own)
}

// csrc/ttnet_split.hip
size_t split_ws_bytes();
int split_pack(const tt_mlp_weights *w, bool critic, void *ws, long long *bump, const RingCursor &cur, hipStream_t stream);
int split_pack_and_sample(const tt_mlp_weights *w, bool critic, void *ws, const RingSample &R, const RingCursor &cur,
                          hipStream_t stream);
int split_forward(bool critic, int n, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                  const ActArgs &act, hipStream_t stream);
#ifdef TT_STAMPS
int split_debug_stamps(unsigned long long *out16);
int split_debug_tiles(unsigned long long *out4096);
int split_debug_poll(unsigned long long *out4);
int split_debug_log(unsigned long long *out, int reset);
int split_debug_block_stamps(unsigned long long *out, int nblocks);
#endif

}  // namespace ttnet
