// ttenv.hip -- libttenv.so: batched truck-trailer backing environment for MI355X (gfx950).
//
// One thread per env, one fused kernel per vector step:
//   clip -> one Dormand-Prince step of the 6-state kinematic ODE (f64) -> 23-dim observation
//   -> reward_functionv1 reward with its carry -> termination flags -> (optional) in-kernel reset.
//
// Memory layout (DESIGN.md "Data layout"): the per-step state is an array of 64-env TILES, one tile per
// wavefront: 14 rows of 64 x 8 B (6 kinematic f64, 6 reward-carry f64, the start distance, and one row of
// {prev_steer f32, packed counters u32}) = 7168 contiguous bytes that the wave reads and writes with
// 512-byte row accesses at immediate offsets from a single base address.  Attributes that only the reset /
// pose-override path touches (start pose, per-env goal, per-env trailer length) live in separate SoA rows.
// The observation tile of a workgroup is staged through LDS so that each wave stores whole 16-byte vectors
// of the row-major [N,23] f32 matrix instead of 64 words 92 bytes apart.
//
// What is restated from the reference (paths relative to pain7576/ddpg-trucktrailer):
//   kinematic ODE            truck_trailer_sim/simv2.py:269-303
//   observation              truck_trailer_sim/simv2.py:103-181
//   reset / pose override    truck_trailer_sim/simv2.py:459-498, 263-267; DDPG/test.py:96-115
//   step, flags, done        truck_trailer_sim/simv2.py:499-545, 305-345
//   reward + carry           truck_trailer_sim/reward_functionv1.py:6-109, 144-506
//   integrator               scipy RK45 tableau (scipy/integrate/_ivp/rk.py), ONE step of h = dt
//
// Arithmetic shortcuts taken here (the CPU oracle takes none, so the parity tests check them):
//   * only the two headings feed back into the ODE's right-hand side, and they move by < 0.1 rad within a
//     step, so each stage's sin/cos is a small-angle rotation of the step's initial sin/cos (3 full-range
//     sincos per step instead of 18);
//   * the hitch-angle and obs[19..22] sin/cos come from angle-difference identities and (dx, dy)/distance;
//   * the three angles the reward reads back through float32 (np.arctan2 of f32 sin/cos) are reproduced
//     by their exactly-rounded value (float)(a + e), e the first-order effect of the f32 rounding;
//   * tanh(x) = (1 - t)/(1 + t), t = exp(-2|x|); cos(wrap(x)) = cos(x).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ttenv.h"

#ifndef TT_TABLE
#define TT_TABLE 0  // 1: polynomial / tableau constants come from the kernarg table instead of literals
#endif

namespace {

constexpr int OBS = TT_OBS_DIM;
#ifndef TT_BLOCK
#define TT_BLOCK 256
#endif
constexpr int BLOCK = TT_BLOCK;  // threads per workgroup (a multiple of the 64-env tile)
constexpr int TILE = 64;
constexpr double kPi = 3.14159265358979323846;
constexpr double kDeg = kPi / 180.0;

// rows of a hot tile [H_ROWS][TILE] (8 bytes per entry)
enum HotRow : int {
    H_PSI1 = 0, H_PSI2, H_X1, H_Y1, H_X2, H_Y2,      // kinematic state
    H_D3, H_D2, H_D1, H_PREV, H_CUM, H_CLOSEST,      // reward carry: distance window, backward sum, closest
    H_DINIT,                                         // |goal - start| (without the 1e-6 of reward_functionv1.py:35)
    H_MISC,                                          // {prev_steer f32, packed counters u32}
    H_ROWS
};
// cold SoA rows [C_ROWS][npad]
enum ColdRow : int { C_SX = 0, C_SY, C_SYAW, C_GX, C_GY, C_GYAW, C_SG, C_CG, C_L2, C_ROWS };

// packed per-env counters: steps [0,12) | max_episode_steps [12,24) | stage latches 2, 3 [24,26) | carry age [26,32)
//   steps      env.episode_steps (simv2.py:523-525): what the exploration tiers, the max-step penalty and the max-steps flag read;
//              a caller may write it (tt_env_set_steps; episode_replay_collectorv2.py:269) WITHOUT touching the reward carry;
//   stages     stages_achieved[1], [2] of the reward carry (reward_functionv1.py:338-367); [0] is written but never read (Q2:
//              the stage-1 bonus is paid on every step inside 5 m), so it is not kept;
//   carry age  step_count_for_backward_tracking of the reward carry (:57-60, :258), saturating at 63 -- only min(1, n / 50) is
//              read (:263) -- and 0 <=> `reward_state is None` (simv2.py:349): the next step builds the carry afresh.
__host__ __device__ inline uint32_t pk_steps(uint32_t p) { return p & 0xFFFu; }
__host__ __device__ inline uint32_t pk_max(uint32_t p) { return (p >> 12) & 0xFFFu; }
__host__ __device__ inline uint32_t pk_stages(uint32_t p) { return (p >> 24) & 0x3u; }
__host__ __device__ inline uint32_t pk_age(uint32_t p) { return p >> 26; }
__host__ __device__ inline uint32_t pk_make(uint32_t steps, uint32_t maxs, uint32_t stages, uint32_t age) {
    return (steps > 0xFFFu ? 0xFFFu : steps) | ((maxs > 0xFFFu ? 0xFFFu : maxs) << 12) | ((stages & 3u) << 24) |
           ((age > 63u ? 63u : age) << 26);
}

// numeric constants of the kernel's polynomials and of the RK45 tableau
struct KTable {
    double f[17];     // 1/k!, k = 0..16
    double A[6][5];   // scipy RK45.A
    double B[6];      // scipy RK45.B (5th-order weights)
    double C[6];      // scipy RK45.C
    double two_over_pi, p1, p2, p3, p3t;  // pi/2 in 33-bit pieces (Cody-Waite)
    double inv_2pi, tp_hi, tp_lo;         // 2*pi split
    double log2e, ln2_hi, ln2_lo;
};

constexpr double fact_inv(int k) {
    double r = 1.0;
    for (int i = 2; i <= k; ++i) r /= (double)i;
    return r;
}

constexpr KTable make_table() {
    KTable t{};
    for (int k = 0; k <= 16; ++k) t.f[k] = fact_inv(k);
    constexpr double A[6][5] = {{0, 0, 0, 0, 0},
                                {1.0 / 5, 0, 0, 0, 0},
                                {3.0 / 40, 9.0 / 40, 0, 0, 0},
                                {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                                {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                                {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
    constexpr double B[6] = {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
    constexpr double C[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 5; ++j) t.A[i][j] = A[i][j];
        t.B[i] = B[i];
        t.C[i] = C[i];
    }
    t.two_over_pi = 0.6366197723675814;
    t.p1 = 0x1.921fb54400000p+0; t.p2 = 0x1.0b4611a600000p-34; t.p3 = 0x1.3198a2e000000p-69; t.p3t = 0x1.b839a252049c1p-104;
    t.inv_2pi = 0.15915494309189535; t.tp_hi = 0x1.921fb54442000p+2; t.tp_lo = 2.9774189921946493e-12;
    t.log2e = 1.4426950408889634; t.ln2_hi = 0x1.62e42fee00000p-1; t.ln2_lo = 0x1.a39ef35793c76p-33;
    return t;
}

struct KParams {
    double v, v_over_L1, ho, h;
    double L2;
    double gx, gy, gyaw, sg, cg;
    double minx, maxx, miny, maxy;
    double cx, cy, inv_hx, inv_hy, inv_M;
    double max_steer, pos_thr, ori_thr, step_length, inv_step_length;
    double rlo[3], rhi[3];
    int extra_steps, fixed_max;
    unsigned term_mask;
    int npad;  // envs rounded up to a whole tile: row stride of the cold block
    int stateless;       // simv1.py:435: the reward carries nothing from step to step
    int pool_m;          // > 0: resets draw from pool[pool_m][3] instead of the box rlo..rhi
    int nt;              // 1: obs / reward / done leave as non-temporal stores (set by N: see nt_stores_for)
    const double *pool;
#if TT_TABLE
    KTable t;
#endif
};

#if TT_TABLE
#define TT_T(P) ((P).t)
#else
__device__ constexpr KTable kTable = make_table();
#define TT_T(P) kTable
#endif

struct Info {
    double *comp;
    uint8_t *violation;
    uint8_t *flags;
};

struct Bufs {
    double *hot;         // [ntiles][H_ROWS][TILE]
    double *cold;        // [C_ROWS][npad]
    uint32_t *episodes;  // [npad] episode number of each env (keys the reset RNG)
    long long *counter;  // optional caller-owned count of vector steps (tt_env_set_step_counter), advanced by k_step
};

__device__ __forceinline__ double *hot_ptr(double *hot, int i) {
    return hot + (size_t)(i >> 6) * (H_ROWS * TILE) + (i & (TILE - 1));
}
__device__ __forceinline__ const double *hot_ptr(const double *hot, int i) {
    return hot + (size_t)(i >> 6) * (H_ROWS * TILE) + (i & (TILE - 1));
}

// ------------------------------------------------------------------------------------------
// counter-based RNG (Philox4x32-10, Salmon et al. 2011)
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
__device__ inline double u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// ------------------------------------------------------------------------------------------
// f64 elementary functions sized for this kernel: branch-free, Taylor coefficients 1/k! shared by all of
// them (sin/cos are written in u = -x^2 so that every coefficient is positive).

// sin and cos of any heading with |x| < 1e6: Cody-Waite reduction by pi/2 split into 33-bit pieces
// (k*P1 and k*P2 are exact for |k| < 2^20), then Taylor on [-pi/4, pi/4] (sin to x^15, cos to x^16: < 5e-17).
__device__ __forceinline__ void tt_sincos(const KTable &T, double x, double &s, double &c) {
    const double k = rint(x * T.two_over_pi);
    double r = fma(-k, T.p1, x);
    r = fma(-k, T.p2, r);
    r = fma(-k, T.p3, r);
    r = fma(-k, T.p3t, r);
    const double u = -(r * r);
    double ps = T.f[15];
    ps = fma(ps, u, T.f[13]);
    ps = fma(ps, u, T.f[11]);
    ps = fma(ps, u, T.f[9]);
    ps = fma(ps, u, T.f[7]);
    ps = fma(ps, u, T.f[5]);
    ps = fma(ps, u, T.f[3]);
    const double sr = fma(r * u, ps, r);
    double pc = T.f[16];
    pc = fma(pc, u, T.f[14]);
    pc = fma(pc, u, T.f[12]);
    pc = fma(pc, u, T.f[10]);
    pc = fma(pc, u, T.f[8]);
    pc = fma(pc, u, T.f[6]);
    pc = fma(pc, u, T.f[4]);
    pc = fma(pc, u, T.f[2]);
    const double cr = fma(u, pc, 1.0);
    const int q = (int)(long long)k;
    const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// sin and cos of a small angle, no reduction: the heading increments inside one 0.08 s step, bounded by
// dt*|v|/L (0.08 rad for the reference constants; tt_env_create refuses > 0.25).  sin to x^11, cos to x^12:
// truncation 1e-19 at 0.08 rad, 1e-16 at 0.25 rad.
__device__ __forceinline__ void tt_sincos_small(const KTable &T, double x, double &s, double &c) {
    const double u = -(x * x);
    double ps = T.f[11];
    ps = fma(ps, u, T.f[9]);
    ps = fma(ps, u, T.f[7]);
    ps = fma(ps, u, T.f[5]);
    ps = fma(ps, u, T.f[3]);
    s = fma(x * u, ps, x);
    double pc = T.f[12];
    pc = fma(pc, u, T.f[10]);
    pc = fma(pc, u, T.f[8]);
    pc = fma(pc, u, T.f[6]);
    pc = fma(pc, u, T.f[4]);
    pc = fma(pc, u, T.f[2]);
    c = fma(u, pc, 1.0);
}

// exp(y) for y <= 0 (clamped at -100): k = rint(y/ln2), Taylor to r^12 on |r| <= ln2/2 (truncation 2e-16)
__device__ __forceinline__ double tt_exp_neg(const KTable &T, double y) {
    y = fmax(y, -100.0);
    const double k = rint(y * T.log2e);
    double r = fma(-k, T.ln2_hi, y);
    r = fma(-k, T.ln2_lo, r);
    double p = T.f[12];
    p = fma(p, r, T.f[11]);
    p = fma(p, r, T.f[10]);
    p = fma(p, r, T.f[9]);
    p = fma(p, r, T.f[8]);
    p = fma(p, r, T.f[7]);
    p = fma(p, r, T.f[6]);
    p = fma(p, r, T.f[5]);
    p = fma(p, r, T.f[4]);
    p = fma(p, r, T.f[3]);
    p = fma(p, r, T.f[2]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// The reference reads three angles back through float32: np.arctan2(f32(sin a), f32(cos a)).  Its real value
// is a + e with e = (y*cos a - x*sin a)/(x*cos a + y*sin a) (y, x the f32-rounded sin, cos; |e| ~ 1e-8, the
// first-order form is exact to 1e-21), so the correctly rounded f32 result is (float)(a + e): no atan needed.
// `a` must already be wrapped to [-pi, pi].
__device__ __forceinline__ float tt_atan2_readback(double a, double sa, double ca, float y, float x) {
    const double yd = (double)y, xd = (double)x;
    const double num = yd * ca - xd * sa, den = xd * ca + yd * sa;  // den = 1 - O(1e-8)
    return (float)(a + num * (2.0 - den));
}

__device__ __forceinline__ double tt_wrap_pi(const KTable &T, double a) {
    const double k = rint(a * T.inv_2pi);
    return fma(-k, T.tp_lo, fma(-k, T.tp_hi, a));
}

// ------------------------------------------------------------------------------------------
struct Goal {
    double gx, gy, sg, cg;
};

// 23-dim observation in f64 from the state and the sin/cos the caller already has (simv2.py:103-181).
__device__ __forceinline__ void observe(const KParams &P, const Goal &g, double x1, double y1, double x2, double y2,
                                        double s1, double c1, double s2, double c2, double sd, double cd, double dx,
                                        double dy, double cur, double inv_cur, float *of) {
    const double dxl = dx * c2 + dy * s2, dyl = -dx * s2 + dy * c2;
    of[0] = (float)((x1 - P.cx) * P.inv_hx); of[1] = (float)((y1 - P.cy) * P.inv_hy);
    of[2] = (float)s1; of[3] = (float)c1;
    of[4] = (float)((x2 - P.cx) * P.inv_hx); of[5] = (float)((y2 - P.cy) * P.inv_hy);
    of[6] = (float)s2; of[7] = (float)c2;
    of[8] = (float)(s1 * c2 - c1 * s2); of[9] = (float)(c1 * c2 + s1 * s2);
    of[10] = (float)sd; of[11] = (float)cd;
    of[12] = (float)((g.gx - P.cx) * P.inv_hx); of[13] = (float)((g.gy - P.cy) * P.inv_hy);
    of[14] = (float)g.sg; of[15] = (float)g.cg;
    of[16] = (float)fmin(fmax(cur * P.inv_M, 0.0), 1.0);
    of[17] = (float)fmin(fmax(dxl * P.inv_M, -1.0), 1.0);
    of[18] = (float)fmin(fmax(dyl * P.inv_M, -1.0), 1.0);
    of[19] = (float)(g.sg * c2 - g.cg * s2); of[20] = (float)(g.cg * c2 + g.sg * s2);
    // sin/cos of atan2(dy,dx) - (psi2 + pi); atan2(0,0) = 0
    of[21] = (float)(cur > 0.0 ? -dyl * inv_cur : s2);
    of[22] = (float)(cur > 0.0 ? -dxl * inv_cur : -c2);
}

// Store a workgroup's [nv,23] f32 observation tile, staged through LDS so the global stores are
// contiguous 16-byte vectors.  Every thread of the block must call this (barriers inside).
// nt: non-temporal stores -- at env counts whose per-step traffic is far beyond the caches, the 97 B per env-step that nobody
// re-reads before they are evicted anyway (observation, reward, done: ~half of all written bytes) stop displacing the state
// tiles on their way out: 278 -> 248 us per launch at 4 M envs (0.59 -> 0.66 of 8 TB/s), nothing at 65,536 / 1 M envs, where
// the policy's next launch finds the observations in L2 / the Infinity Cache (profiles/r03_nsweep.md).
__device__ inline void store_obs_tile(float *tile, const float *of, bool valid, float *obs, int block_first, int nv,
                                      const bool nt = false) {
    const int tid = threadIdx.x;
    if (valid) {
#pragma unroll
        for (int j = 0; j < OBS; ++j) tile[tid * OBS + j] = of[j];  // stride 23 words: conflict-free
    }
    __syncthreads();
    float *dst = obs + (size_t)block_first * OBS;
    const int total = nv * OBS;
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int nvec = total >> 2;
        const float4 *t4 = reinterpret_cast<const float4 *>(tile);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        if (nt) {
            using v4f = __attribute__((ext_vector_type(4))) float;
            for (int q = tid; q < nvec; q += BLOCK) {
                const float4 t = t4[q];
                __builtin_nontemporal_store(v4f{t.x, t.y, t.z, t.w}, reinterpret_cast<v4f *>(d4 + q));
            }
        } else {
            for (int q = tid; q < nvec; q += BLOCK) d4[q] = t4[q];
        }
        for (int q = (nvec << 2) + tid; q < total; q += BLOCK) dst[q] = tile[q];
    } else {
        for (int q = tid; q < total; q += BLOCK) dst[q] = tile[q];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// one env in registers
struct Env {
    double psi1, psi2, x1, y1, x2, y2;      // kinematic state
    double d3, d2, d1, prev, cum, closest;  // reward carry
    double dinit;
    float psteer;
    uint32_t pk;
    Goal g;
    double gyaw, L2;
};

template <bool PER_ENV>
__device__ __forceinline__ void load_env(const KParams &P, const Bufs &b, int i, Env &e) {
    const double *h = hot_ptr(b.hot, i);
    e.psi1 = h[H_PSI1 * TILE]; e.psi2 = h[H_PSI2 * TILE];
    e.x1 = h[H_X1 * TILE]; e.y1 = h[H_Y1 * TILE]; e.x2 = h[H_X2 * TILE]; e.y2 = h[H_Y2 * TILE];
    e.d3 = h[H_D3 * TILE]; e.d2 = h[H_D2 * TILE]; e.d1 = h[H_D1 * TILE]; e.prev = h[H_PREV * TILE];
    e.cum = h[H_CUM * TILE]; e.closest = h[H_CLOSEST * TILE];
    e.dinit = h[H_DINIT * TILE];
    const uint2 m = *reinterpret_cast<const uint2 *>(h + H_MISC * TILE);
    e.psteer = __uint_as_float(m.x);
    e.pk = m.y;
    if (PER_ENV) {
        const double *c = b.cold + i;
        const size_t S = (size_t)P.npad;
        e.g.gx = c[C_GX * S]; e.g.gy = c[C_GY * S]; e.g.sg = c[C_SG * S]; e.g.cg = c[C_CG * S];
        e.gyaw = c[C_GYAW * S]; e.L2 = c[C_L2 * S];
    } else {
        e.g.gx = P.gx; e.g.gy = P.gy; e.g.sg = P.sg; e.g.cg = P.cg;
        e.gyaw = P.gyaw; e.L2 = P.L2;
    }
}

__device__ __forceinline__ void store_env(const Bufs &b, int i, const Env &e) {
    double *h = hot_ptr(b.hot, i);
    h[H_PSI1 * TILE] = e.psi1; h[H_PSI2 * TILE] = e.psi2;
    h[H_X1 * TILE] = e.x1; h[H_Y1 * TILE] = e.y1; h[H_X2 * TILE] = e.x2; h[H_Y2 * TILE] = e.y2;
    h[H_D3 * TILE] = e.d3; h[H_D2 * TILE] = e.d2; h[H_D1 * TILE] = e.d1; h[H_PREV * TILE] = e.prev;
    h[H_CUM * TILE] = e.cum; h[H_CLOSEST * TILE] = e.closest;
    *reinterpret_cast<uint2 *>(h + H_MISC * TILE) = make_uint2(__float_as_uint(e.psteer), e.pk);
}

// Place an env (in registers) at a start pose (simv2.py:481-496 / DDPG/test.py:96-115): trailer at the pose,
// truck L2 ahead, state rounded to float32 (simv2.py:489), episode cleared.  Writes the cold attributes.
// `of` (may be NULL) receives the first observation (steering 0).
__device__ inline void place_env(const KParams &P, const Bufs &b, int i, Env &e, double sx, double sy, double syaw,
                                 const Goal &g, double gyaw, double L2, float *of) {
    const KTable &T = TT_T(P);
    double ss, cs;
    tt_sincos(T, syaw, ss, cs);
    e.psi1 = e.psi2 = (double)(float)syaw;
    e.x1 = (double)(float)(sx + L2 * cs);
    e.y1 = (double)(float)(sy + L2 * ss);
    e.x2 = (double)(float)sx;
    e.y2 = (double)(float)sy;
    const double ddx = g.gx - sx, ddy = g.gy - sy;
    e.dinit = sqrt(ddx * ddx + ddy * ddy);
    e.g = g; e.gyaw = gyaw; e.L2 = L2;
    const int maxs = P.fixed_max > 0 ? P.fixed_max : (int)(e.dinit / P.step_length) + P.extra_steps;
    e.pk = pk_make(0u, (uint32_t)maxs, 0u, 0u);
    // carry rows are re-initialised by the first step (carry age 0); keep them finite
    e.d3 = e.d2 = e.d1 = e.prev = e.closest = e.dinit; e.cum = 0.0; e.psteer = 0.0f;
    hot_ptr(b.hot, i)[H_DINIT * TILE] = e.dinit;  // read-only for the step kernel: store_env() does not write it
    double *c = b.cold + i;
    const size_t S = (size_t)P.npad;
    c[C_SX * S] = sx; c[C_SY * S] = sy; c[C_SYAW * S] = syaw;
    c[C_GX * S] = g.gx; c[C_GY * S] = g.gy; c[C_GYAW * S] = gyaw; c[C_SG * S] = g.sg; c[C_CG * S] = g.cg;
    c[C_L2 * S] = L2;
    if (of) {
        double sp, cp;
        tt_sincos(T, e.psi1, sp, cp);
        const double dx = g.gx - e.x2, dy = g.gy - e.y2;
        const double cur = sqrt(dx * dx + dy * dy);
        observe(P, g, e.x1, e.y1, e.x2, e.y2, sp, cp, sp, cp, 0.0, 1.0, dx, dy, cur, cur > 0.0 ? 1.0 / cur : 0.0, of);
    }
}

// start pose of episode number `episode` of env `env` under reset seed `seed`: a pure function of the three
__device__ inline void random_pose(const KParams &P, uint64_t seed, uint32_t env, uint32_t episode, double &sx, double &sy,
                                   double &syaw) {
    uint32_t r[4];
    philox4x32(env, episode, 0u, 0x7452u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    if (P.pool_m > 0) {  // pre-filtered poses (simv1's Dubins-feasible starts)
        const double *q = P.pool + 3 * (size_t)(r[3] % (uint32_t)P.pool_m);
        sx = q[0]; sy = q[1]; syaw = q[2];
        return;
    }
    sx = P.rlo[0] + (P.rhi[0] - P.rlo[0]) * u01(r[0]);  // draw order x, y, yaw (simv2.py:331-333)
    sy = P.rlo[1] + (P.rhi[1] - P.rlo[1]) * u01(r[1]);
    syaw = P.rlo[2] + (P.rhi[2] - P.rlo[2]) * u01(r[2]);
}

__device__ __forceinline__ float random_action(uint64_t seed, uint32_t env, uint32_t steps, uint32_t episode) {
    uint32_t r[4];
    philox4x32(env, steps, episode, 0xAC71u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (float)((2.0 * u01(r[0]) - 1.0) * (kPi / 4));
}

struct StepOut {
    double total, c_prog, c_head, c_orient, staged, safety, explore, final_bonus, back, c_smooth, budget;
    uint32_t viol, flags;
    bool done;
};

// env.step for one env held in registers (simv2.py:499-545 + reward_functionv1.py:442-506)
__device__ __forceinline__ void step_env(const KParams &P, Env &e, float action, float *of, StepOut &o) {
    const KTable &T = TT_T(P);
    // ---- simv2.py:504-505: clip in f64 against np.radians(45)
    const double delta = fmin(fmax((double)action, -P.max_steer), P.max_steer);
    double sd, cd;
    tt_sincos(T, delta, sd, cd);
    const double w1 = P.v_over_L1 * (sd / cd);  // truck yaw rate, constant over the step

    // ---- one Dormand-Prince step (scipy RK45 tableau).  Positions do not feed back, so they are accumulated
    // straight into their B-weighted sums; every stage's sin/cos is a small rotation of the initial ones.
    const double h = P.h, v = P.v, vL2 = P.v / e.L2, hoL2 = P.ho / e.L2, how1 = P.ho * w1;
    double sp1, cp1, sp2, cp2;
    tt_sincos(T, e.psi1, sp1, cp1);
    tt_sincos(T, e.psi2, sp2, cp2);
    const double beta = h * w1;
    double k2[6];
    double ax1 = 0.0, ay1 = 0.0, ax2 = 0.0, ay2 = 0.0, apsi2 = 0.0;
    double s1 = sp1, c1 = cp1;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        double s2 = sp2, c2 = cp2;
        if (s > 0) {
            double acc = T.A[s][0] * k2[0];
#pragma unroll
            for (int j = 1; j < s; ++j) acc = fma(T.A[s][j], k2[j], acc);
            double sb, cb, sg2, cg2;
            tt_sincos_small(T, T.C[s] * beta, sb, cb);
            tt_sincos_small(T, h * acc, sg2, cg2);
            s1 = sp1 * cb + cp1 * sb; c1 = cp1 * cb - sp1 * sb;
            s2 = sp2 * cg2 + cp2 * sg2; c2 = cp2 * cg2 - sp2 * sg2;
        }
        const double sth = s1 * c2 - c1 * s2, cth = c1 * c2 + s1 * s2;
        const double v2 = v * cth + how1 * sth;
        k2[s] = vL2 * sth - hoL2 * w1 * cth;
        if (s != 1) {  // B[1] = 0
            const double bs = T.B[s];
            apsi2 = fma(bs, k2[s], apsi2);
            ax1 = fma(bs, v * c1, ax1);
            ay1 = fma(bs, v * s1, ay1);
            ax2 = fma(bs, v2 * c2, ax2);
            ay2 = fma(bs, v2 * s2, ay2);
        }
    }
    // stage 6 sits at c = 1: its truck heading IS the new truck heading, so (s1, c1) carry over
    e.psi1 += beta;
    const double dpsi2 = h * apsi2;
    e.psi2 += dpsi2;
    e.x1 = fma(h, ax1, e.x1); e.y1 = fma(h, ay1, e.y1); e.x2 = fma(h, ax2, e.x2); e.y2 = fma(h, ay2, e.y2);
    double s2, c2;
    {
        double sg2, cg2;
        tt_sincos_small(T, dpsi2, sg2, cg2);
        s2 = sp2 * cg2 + cp2 * sg2; c2 = cp2 * cg2 - sp2 * sg2;
    }

    // ---- observation (simv2.py:519), cast to f32 like np.array(..., dtype=float32)
    const double dx = e.g.gx - e.x2, dy = e.g.gy - e.y2;
    const double cur = sqrt(dx * dx + dy * dy);
    const uint32_t pk = e.pk;
    const uint32_t steps = pk_steps(pk) + 1u;  // episode_steps is incremented before the reward (simv2.py:523)
    uint32_t stages = pk_stages(pk);
    const double init = e.dinit + 1e-6;

    // The reward's three tanh -- dynamic weights (:189-238): tanh(7(jp - 0.3)); progress (:144-187):
    // tanh(prev - cur), tanh((hist[0] - cur)/2) -- are (1 - t)/(1 + t), t = exp(-2|x|).  Two of them, 1/cur
    // and 1/init share ONE division through the product of the denominators.
    const bool first = pk_age(pk) == 0u || P.stateless != 0;  // reward_state is None (:40-76)
    const uint32_t age = first ? 1u : pk_age(pk) + 1u;         // step_count_for_backward_tracking after this step (:258)
    const double prev = first ? cur : e.prev, d3 = first ? cur : e.d3, d1 = first ? cur : e.d1;
    const double inst = prev - cur;
    const double net = (d3 - cur) * 0.5;
    const double t_i = tt_exp_neg(T, -2.0 * fabs(inst)), t_n = tt_exp_neg(T, -2.0 * fabs(net));
    const double q_i = 1.0 + t_i, q_n = 1.0 + t_n;
    const double cur_s = cur > 0.0 ? cur : 1.0;
    const double qq = q_i * q_n, ci = cur_s * init;
    const double r_all = 1.0 / (qq * ci);
    const double inv_cur = cur > 0.0 ? r_all * (qq * init) : 0.0;
    const double inv_init = r_all * (qq * cur_s);
    const double th = copysign((1.0 - t_i) * (r_all * (ci * q_n)), inst);
    const double tn = copysign((1.0 - t_n) * (r_all * (ci * q_i)), net);

    observe(P, e.g, e.x1, e.y1, e.x2, e.y2, s1, c1, s2, c2, sd, cd, dx, dy, cur, inv_cur, of);

    // ---- reward (reward_functionv1.py:442-506)
    // int(initial_distance / 0.40096) + 75 (:38): the quotient by one multiplication, and by the real division (what
    // numpy does) only where the two could truncate differently -- next to an integer (wave-uniform, almost never)
    const double rq = init * P.inv_step_length;
    int rmax = (int)rq;
    if (__any(fabs(rq - rint(rq)) <= 1e-11 * rq)) rmax = (int)(init / P.step_length);
    rmax += P.extra_steps;
    const float steer_now = tt_atan2_readback(delta, sd, cd, of[10], of[11]);  // np.arctan2(obs[10], obs[11]) (:37)
    if (first) {
        e.cum = 0.0;
        e.closest = cur;
        e.psteer = steer_now;
        stages = 0u;
    } else if (cur < e.closest) {
        e.closest = cur;
    }
    const double jp = fmin(fmax((init - cur) * inv_init, 0.0), 1.0);
    const double wa = 7.0 * (jp - 0.3);
    const double t_w = tt_exp_neg(T, -2.0 * fabs(wa));
    const double tw = copysign((1.0 - t_w) / (1.0 + t_w), wa);
    const double w_orient = (tw + 1.0) * 0.5;
    const double w_head = 1.0 - w_orient;
    const bool mono = (d1 >= prev) && (prev >= cur);
    const double progress = (inst > 0.0 ? th : th * 0.5) + tn * 0.5 + (mono ? 0.2 : 0.0);
    // heading (:285-309): cos(atan2(dy,dx) - (co + pi)) with co = np.arctan2(obs[6], obs[7]) in f32.
    // co = wrap(psi2) + eps with |eps| ~ 1e-7, so its sin/cos are a rotation of (s2, c2) by eps.
    double heading;
    {
        const double w2 = tt_wrap_pi(T, e.psi2);
        const double eps = (double)tt_atan2_readback(w2, s2, c2, of[6], of[7]) - w2;
        const double half = 1.0 - 0.5 * eps * eps;
        const double sco = s2 * half + c2 * eps, cco = c2 * half - s2 * eps;
        heading = cur > 0.0 ? -(dx * cco + dy * sco) * inv_cur : -cco;
    }
    // orientation (:311-324): f32 * 15.0 stays f32 in numpy
    const float orient15 = of[20] * 15.0f;
    // staged bonuses (:338-367) and success: only within 5 m of the goal (wave-uniform skip otherwise)
    double staged = 0.0;
    bool at_goal = false;
    if (__any(cur <= 5.0)) {
        // |np.arctan2(obs[19], obs[20])| with obs[19], obs[20] = sin, cos(goalyaw - psi2)
        const double oa = tt_wrap_pi(T, e.gyaw - e.psi2);
        const double ori_err = (double)fabsf(tt_atan2_readback(oa, e.g.sg * c2 - e.g.cg * s2, e.g.cg * c2 + e.g.sg * s2,
                                                               of[19], of[20]));
        if (cur <= 5.0) staged += 10.0;   // every step, no latch (Q2)
        if (cur <= 2.0 && ori_err <= 45.0 * kDeg && !(stages & 1u)) { staged += 25.0; stages |= 1u; }
        at_goal = cur <= P.pos_thr && ori_err <= P.ori_thr;
        if (at_goal && !(stages & 2u)) { staged += 100.0; stages |= 2u; }
    }
    // safety (:369-421)
    double safety = 0.0;
    uint32_t viol = TT_V_NONE;
    const double hitch = fabs(e.psi1 - e.psi2);
    if (hitch > 85.0 * kDeg) { safety += -500.0; viol = TT_V_JACKKNIFE; }
    else if (hitch > 70.0 * kDeg) { safety += -50.0; viol = TT_V_JACKKNIFE_WARNING; }
    const double lox = fmin(e.x1, e.x2), hix = fmax(e.x1, e.x2), loy = fmin(e.y1, e.y2), hiy = fmax(e.y1, e.y2);
    const bool outside = lox < P.minx || hix > P.maxx || loy < P.miny || hiy > P.maxy;
    if (lox < P.minx - 2.0 || hix > P.maxx + 2.0 || loy < P.miny - 2.0 || hiy > P.maxy + 2.0) {
        safety += -500.0; viol = TT_V_MAJOR_BOUNDARY;
    } else if (outside) {
        safety += -50.0; viol = TT_V_MINOR_BOUNDARY;
    }
    const bool passed = e.g.gy > e.y2;
    if (passed) { safety += -500.0; viol = TT_V_PAST_THE_GOAL; }
    if ((int)steps >= rmax) { safety += -500.0; viol = TT_V_MAX_STEP; }
    const bool excessive = cur > e.closest + 6.0;  // :120-124
    if (excessive) { safety += -500.0; viol = TT_V_EXCESSIVE_BACKWARD; }
    // exploration (:423-439)
    const double explore = (double)steps < rmax * 0.5 ? 4.0 : ((double)steps < rmax * 0.8 ? 2.0 : 0.0);
    // backward-movement budget (:240-283)
    e.cum += fmax(0.0, cur - prev);
    const double budget = 5.0 * fmin(1.0, (double)age * 0.02);  // step_count_for_backward_tracking (stateless: always 1)
    const double excess = fmax(0.0, e.cum - budget);
    double back = 0.0;
    if (__any(excess > 0.0)) back = excess > 0.0 ? -(excess * sqrt(excess)) * 0.5 : 0.0;
    // smoothness against the episode's FIRST steering (:326-335; previous_steering is never refreshed)
    const double smooth = (double)fabsf(steer_now - e.psteer) * (1.0 / (90.0 * kDeg));
    o.final_bonus = at_goal ? 200.0 : 0.0;

    o.c_prog = progress * 15.0; o.c_head = heading * 15.0 * w_head;
    o.c_orient = (double)orient15 * w_orient; o.c_smooth = smooth * -25.0;
    o.staged = staged; o.safety = safety; o.explore = explore; o.back = back; o.budget = budget;
    o.total = 0.0 + o.c_prog + o.c_head + o.c_orient + staged + safety + explore + back + o.c_smooth + o.final_bonus;

    // ---- flags (simv2.py:528-541)
    uint32_t fl = 0u;
    if (hitch > 90.0 * kDeg) fl |= TT_F_JACKKNIFE;
    if (outside) fl |= TT_F_OUT_OF_MAP;
    if (steps >= pk_max(pk)) fl |= TT_F_MAX_STEPS;
    if (at_goal) fl |= TT_F_GOAL_REACHED | TT_F_SUCCESS;
    if (passed) fl |= TT_F_GOAL_PASSED;
    if (excessive) fl |= TT_F_EXCESSIVE_BACK;
    o.flags = fl; o.viol = viol;
    o.done = (fl & P.term_mask) != 0u;

    // carry for the next step: window [hist1..hist4] <- [hist2, hist3, hist4, cur]
    e.d3 = first ? cur : e.d2; e.d2 = d1; e.d1 = prev; e.prev = cur;
    e.pk = pk_make(steps, pk_max(pk), stages, age);
}

__device__ __forceinline__ void write_info(const Info &info, size_t N, int i, const Env &e, const StepOut &o) {
    if (info.comp) {
        double *c = info.comp + i;
        c[TT_I_TOTAL * N] = o.total; c[TT_I_PROGRESS * N] = o.c_prog; c[TT_I_HEADING * N] = o.c_head;
        c[TT_I_ORIENT * N] = o.c_orient; c[TT_I_STAGED * N] = o.staged; c[TT_I_SAFETY * N] = o.safety;
        c[TT_I_EXPLORE * N] = o.explore; c[TT_I_FINAL * N] = o.final_bonus; c[TT_I_BACKWARD * N] = o.back;
        c[TT_I_SMOOTH * N] = o.c_smooth; c[TT_I_CUMBACK * N] = e.cum; c[TT_I_BUDGET * N] = o.budget;
    }
    if (info.violation) info.violation[i] = (uint8_t)o.viol;
    if (info.flags) info.flags[i] = (uint8_t)o.flags;
}

// ------------------------------------------------------------------------------------------
// the vector step.  RANDOM_POLICY: the action is drawn in-kernel (BASELINE.json config 2), keyed by
// (policy seed, env, step-in-episode, episode number) so that a captured hipGraph replays correctly.
#ifdef TT_STAMPS
#include "ttstamps.h"
__device__ unsigned long long g_stepblk[1024][2];
__device__ TTLog g_log_step;
#endif
template <bool PER_ENV, bool INFO, bool AUTO_RESET, bool RANDOM_POLICY>
__global__ __launch_bounds__(BLOCK) void k_step(const KParams P, const int n, const Bufs b,
                                                const float *__restrict__ action, float *__restrict__ action_out,
                                                float *__restrict__ obs, float *__restrict__ reward,
                                                uint8_t *__restrict__ done, const Info info, const uint64_t seed,
                                                const uint64_t policy_seed, const int *__restrict__ cursor) {
    __shared__ __attribute__((aligned(16))) float tile[BLOCK * OBS];
#ifdef TT_STAMPS      // diagnostic build: wall clock (100 MHz) at the begin and end of every workgroup (tools/step_timeline.py)
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_stepblk[blockIdx.x][0] = wall_clock64();
#endif
    if (cursor) {       // ring addressing (tt_env_step_ring): obs / reward / done are the ring's bases, the slots come from the cursor
        obs += (size_t)cursor[1] * n * OBS;
        reward += (size_t)cursor[0] * n;
        done += (size_t)cursor[0] * n;
    }
    const int block_first = blockIdx.x * BLOCK;
    const int i = block_first + threadIdx.x;
    const bool valid = i < n;
    const int nv = min(BLOCK, n - block_first);
    float of[OBS];

    if (b.counter && i == 0) *b.counter += 1;      // one more vector step stored (read by later launches only)
    if (valid) {
        Env e;
        load_env<PER_ENV>(P, b, i, e);
        float a = 0.f;
        if (!RANDOM_POLICY) a = action[i];
        // A lone wave per SIMD hides no latency: every load of the step is requested HERE, in one burst (one trip to
        // L2 / the Infinity Cache instead of the four or five the scheduler otherwise spreads through the arithmetic by
        // sinking each load next to its first use)
        __builtin_amdgcn_sched_barrier(0);
        if (RANDOM_POLICY) {
            a = random_action(policy_seed, (uint32_t)i, pk_steps(e.pk), b.episodes[i]);
            if (action_out) action_out[i] = a;
        }
        StepOut o;
#ifdef TT_DBG_STEP_MEM_ONLY      // diagnostic build (wrong results): the step's memory traffic without its arithmetic
        o = StepOut{};
        o.total = e.psi1 + a; o.done = false;
        e.psi1 += 1e-9; e.psi2 += e.x1 * 1e-12; e.d3 = e.d2; e.d2 = e.d1; e.d1 = e.prev; e.cum += e.closest * 1e-12;
#pragma unroll
        for (int j = 0; j < OBS; ++j) of[j] = (float)(e.x2 + j);
#else
        step_env(P, e, a, of, o);
#endif
        if (P.nt) {
            __builtin_nontemporal_store((float)o.total, reward + i);
            __builtin_nontemporal_store((uint8_t)(o.done ? 1 : 0), done + i);
        } else {
            reward[i] = (float)o.total;
            done[i] = o.done ? 1 : 0;
        }
        if (INFO) write_info(info, (size_t)n, i, e, o);
        if (AUTO_RESET && o.done) {
            const uint32_t ep = b.episodes[i] + 1u;
            b.episodes[i] = ep;
            double sx, sy, syaw;
            random_pose(P, seed, (uint32_t)i, ep, sx, sy, syaw);
            const Goal g0{P.gx, P.gy, P.sg, P.cg};
            place_env(P, b, i, e, sx, sy, syaw, g0, P.gyaw, e.L2, of);
        }
        store_env(b, i, e);
    }
    store_obs_tile(tile, of, valid, obs, block_first, nv, P.nt != 0);
#ifdef TT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        g_stepblk[blockIdx.x][1] = wall_clock64();
        tt_log_add(g_log_step, g_stepblk[blockIdx.x][0], g_stepblk[blockIdx.x][1]);
    }
#endif
}

// K vector steps in ONE launch with the random policy: the env stays in registers, only the last
// observation is stored; per-env reward sums and episode counts are accumulated.
template <bool PER_ENV>
__global__ __launch_bounds__(BLOCK) void k_rollout(const KParams P, const int n, const Bufs b, const int k_steps,
                                                   float *__restrict__ obs, float *__restrict__ reward_sum,
                                                   int32_t *__restrict__ episodes_done, const uint64_t seed,
                                                   const uint64_t policy_seed) {
    __shared__ __attribute__((aligned(16))) float tile[BLOCK * OBS];
    const int block_first = blockIdx.x * BLOCK;
    const int i = block_first + threadIdx.x;
    const bool valid = i < n;
    float of[OBS];
    if (valid) {
        Env e;
        load_env<PER_ENV>(P, b, i, e);
        uint32_t ep = b.episodes[i];
        double rsum = 0.0;
        int ndone = 0;
        for (int t = 0; t < k_steps; ++t) {
            const float a = random_action(policy_seed, (uint32_t)i, pk_steps(e.pk), ep);
            StepOut o;
            step_env(P, e, a, of, o);
            rsum += o.total;
            if (o.done) {
                ep += 1u;
                ndone += 1;
                double sx, sy, syaw;
                random_pose(P, seed, (uint32_t)i, ep, sx, sy, syaw);
                const Goal g0{P.gx, P.gy, P.sg, P.cg};
                place_env(P, b, i, e, sx, sy, syaw, g0, P.gyaw, e.L2, of);
            }
        }
        b.episodes[i] = ep;
        store_env(b, i, e);
        if (reward_sum) reward_sum[i] = (float)rsum;
        if (episodes_done) episodes_done[i] = ndone;
    }
    if (obs) store_obs_tile(tile, of, valid, obs, block_first, min(BLOCK, n - block_first));
}

// ------------------------------------------------------------------------------------------
// reset / pose / state kernels (not hot)
__global__ __launch_bounds__(BLOCK) void k_reset(const KParams P, const int n, const Bufs b, const uint8_t *mask,
                                                 float *obs, const uint64_t seed) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (mask && !mask[i])) return;
    // a full reset restarts every env's episode count, so reset(seed) is a pure function of the seed
    // (as np.random.seed is); a masked reset moves the chosen envs on to their next episode
    const uint32_t ep = mask ? b.episodes[i] + 1u : 0u;
    b.episodes[i] = ep;
    double sx, sy, syaw;
    random_pose(P, seed, (uint32_t)i, ep, sx, sy, syaw);
    float of[OBS];
    const Goal g{P.gx, P.gy, P.sg, P.cg};
    Env e;
    place_env(P, b, i, e, sx, sy, syaw, g, P.gyaw, b.cold[C_L2 * (size_t)P.npad + i], obs ? of : nullptr);
    store_env(b, i, e);
    if (obs)
        for (int j = 0; j < OBS; ++j) obs[(size_t)i * OBS + j] = of[j];
}

__global__ __launch_bounds__(BLOCK) void k_init(const KParams P, const Bufs b) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P.npad) return;  // padding lanes of the last tile get a valid env too
    b.episodes[i] = 0u;
    const Goal g{P.gx, P.gy, P.sg, P.cg};
    Env e;
    place_env(P, b, i, e, P.gx, P.gy + 30.0, kPi / 2, g, P.gyaw, P.L2, nullptr);
    store_env(b, i, e);
}

__global__ __launch_bounds__(BLOCK) void k_set_pose(const KParams P, const int n, const Bufs b, const int32_t *idx,
                                                    const int k, const double *start, const double *goal,
                                                    const double *L2, float *obs) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    const size_t S = (size_t)P.npad;
    const double *c = b.cold + i;
    Goal g;
    double gyaw;
    if (goal) {
        g.gx = goal[3 * j]; g.gy = goal[3 * j + 1]; gyaw = goal[3 * j + 2];
        tt_sincos(TT_T(P), gyaw, g.sg, g.cg);
    } else {
        g.gx = c[C_GX * S]; g.gy = c[C_GY * S]; gyaw = c[C_GYAW * S]; g.sg = c[C_SG * S]; g.cg = c[C_CG * S];
    }
    const double l2 = L2 ? L2[j] : c[C_L2 * S];
    float of[OBS];
    Env e;
    place_env(P, b, i, e, start[3 * j], start[3 * j + 1], start[3 * j + 2], g, gyaw, l2, obs ? of : nullptr);
    store_env(b, i, e);
    if (obs)
        for (int q = 0; q < OBS; ++q) obs[(size_t)i * OBS + q] = of[q];
}

__global__ __launch_bounds__(BLOCK) void k_set_attrs(const KParams P, const int n, const Bufs b, const int32_t *idx,
                                                     const int k, const double *start, const double *goal,
                                                     const double *L2) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    const size_t S = (size_t)P.npad;
    double *c = b.cold + i;
    if (start) {
        c[C_SX * S] = start[3 * j]; c[C_SY * S] = start[3 * j + 1]; c[C_SYAW * S] = start[3 * j + 2];
    }
    if (goal) {
        double sg, cg;
        tt_sincos(TT_T(P), goal[3 * j + 2], sg, cg);
        c[C_GX * S] = goal[3 * j]; c[C_GY * S] = goal[3 * j + 1]; c[C_GYAW * S] = goal[3 * j + 2];
        c[C_SG * S] = sg; c[C_CG * S] = cg;
    }
    if (L2) c[C_L2 * S] = L2[j];
    const double ddx = c[C_GX * S] - c[C_SX * S], ddy = c[C_GY * S] - c[C_SY * S];
    hot_ptr(b.hot, i)[H_DINIT * TILE] = sqrt(ddx * ddx + ddy * ddy);
}

__global__ __launch_bounds__(BLOCK) void k_set_state(const int n, const Bufs b, const int32_t *idx, const int k,
                                                     const double *state) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    double *h = hot_ptr(b.hot, i);
    for (int r = 0; r < 6; ++r) h[r * TILE] = state[6 * j + r];
}

__global__ __launch_bounds__(BLOCK) void k_get_state(const int n, const Bufs b, double *out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double *h = hot_ptr(b.hot, i);
    for (int r = 0; r < 6; ++r) out[r * (size_t)n + i] = h[r * TILE];
}

__global__ __launch_bounds__(BLOCK) void k_set_max_steps(const int n, const Bufs b, const int32_t *idx, const int k,
                                                         const int32_t *maxs) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    uint2 *m = reinterpret_cast<uint2 *>(hot_ptr(b.hot, i) + H_MISC * TILE);
    const uint32_t p = m->y;
    m->y = pk_make(pk_steps(p), (uint32_t)(maxs[j] < 0 ? 0 : maxs[j]), pk_stages(p), pk_age(p));
}

// `env.episode_steps = k` (simv2.py:94; episode_replay_collectorv2.py:269): the step counter alone -- the reward carry, its
// age and the stage latches stay (the reference keeps them in reward_state, which such a write does not touch)
__global__ __launch_bounds__(BLOCK) void k_set_steps(const int n, const Bufs b, const int32_t *idx, const int k,
                                                     const int32_t *steps) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    uint2 *m = reinterpret_cast<uint2 *>(hot_ptr(b.hot, i) + H_MISC * TILE);
    const uint32_t p = m->y;
    m->y = pk_make((uint32_t)(steps[j] < 0 ? 0 : steps[j]), pk_max(p), pk_stages(p), pk_age(p));
}

__global__ __launch_bounds__(BLOCK) void k_get_episode(const KParams P, const int n, const Bufs b, int32_t *steps,
                                                       int32_t *maxs, double *start, double *goal, double *L2) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = reinterpret_cast<const uint2 *>(hot_ptr(b.hot, i) + H_MISC * TILE)->y;
    if (steps) steps[i] = (int32_t)pk_steps(p);
    if (maxs) maxs[i] = (int32_t)pk_max(p);
    const size_t S = (size_t)P.npad, N = (size_t)n;
    const double *c = b.cold + i;
    if (start) { start[i] = c[C_SX * S]; start[N + i] = c[C_SY * S]; start[2 * N + i] = c[C_SYAW * S]; }
    if (goal) { goal[i] = c[C_GX * S]; goal[N + i] = c[C_GY * S]; goal[2 * N + i] = c[C_GYAW * S]; }
    if (L2) L2[i] = c[C_L2 * S];
}

template <bool PER_ENV>
__global__ __launch_bounds__(BLOCK) void k_observe(const KParams P, const int n, const Bufs b, const float *steering,
                                                   float *obs) {
    __shared__ __attribute__((aligned(16))) float tile[BLOCK * OBS];
    const int block_first = blockIdx.x * BLOCK;
    const int i = block_first + threadIdx.x;
    const bool valid = i < n;
    float of[OBS];
    if (valid) {
        const KTable &T = TT_T(P);
        Env e;
        load_env<PER_ENV>(P, b, i, e);
        double s1, c1, s2, c2, sd = 0.0, cd = 1.0;
        tt_sincos(T, e.psi1, s1, c1);
        tt_sincos(T, e.psi2, s2, c2);
        if (steering) tt_sincos(T, (double)steering[i], sd, cd);
        const double dx = e.g.gx - e.x2, dy = e.g.gy - e.y2;
        const double cur = sqrt(dx * dx + dy * dy);
        observe(P, e.g, e.x1, e.y1, e.x2, e.y2, s1, c1, s2, c2, sd, cd, dx, dy, cur, cur > 0.0 ? 1.0 / cur : 0.0, of);
    }
    store_obs_tile(tile, of, valid, obs, block_first, min(BLOCK, n - block_first));
}

__global__ __launch_bounds__(BLOCK) void k_random_actions(const int n, const uint64_t seed, const uint64_t step,
                                                          float *out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t r[4];
    philox4x32((uint32_t)i, (uint32_t)step, (uint32_t)(step >> 32), 0xAC71u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    out[i] = (float)((2.0 * u01(r[0]) - 1.0) * (kPi / 4));
}

char g_err[256] = "";

}  // namespace

// ==========================================================================================
// host side
struct tt_env {
    int n = 0, npad = 0, device = 0;
    tt_params params{};
    KParams kp{};
    Bufs b{nullptr, nullptr, nullptr, nullptr};
    bool per_env = false;
    uint64_t seed = 0;
    // optional per-launch timing of the step kernel (tt_env_profile): event pairs bound to the dispatch
    bool profiling = false;
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t ev_used = 0;
    double prof_ms = 0.0;
    long prof_launches = 0;
    char err[256] = "";
};

namespace {

int fail(tt_env *e, int code, const char *fmt, ...) {
    char *dst = e ? e->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 256, fmt, ap);
    va_end(ap);
    return code;
}

#define TT_HIP(e, call)                                                                      \
    do {                                                                                     \
        hipError_t err_ = (call);                                                            \
        if (err_ != hipSuccess) return fail((e), TT_EHIP, "%s: %s", #call, hipGetErrorString(err_)); \
    } while (0)

// Non-temporal obs / reward / done stores from this many envs on: a step then writes > 200 MB, past every cache level
// (TT_NT_ENVS overrides: 0 = never, 1 = always).
inline int nt_stores_for(int npad) {
    static const long long thr = [] {
        const char *e = std::getenv("TT_NT_ENVS");
        return e ? std::atoll(e) : 2097152ll;
    }();
    return thr > 0 && npad >= thr ? 1 : 0;
}

KParams make_kparams(const tt_params &p, int npad) {
    KParams k{};
    k.v = p.v1x;
    k.v_over_L1 = p.v1x / p.L1;
    k.ho = p.hitch_offset;
    k.h = p.dt;
    k.L2 = p.L2;
    k.gx = p.goal[0]; k.gy = p.goal[1]; k.gyaw = p.goal[2];
    k.sg = std::sin(p.goal[2]); k.cg = std::cos(p.goal[2]);
    k.minx = p.map_min_x; k.maxx = p.map_max_x; k.miny = p.map_min_y; k.maxy = p.map_max_y;
    const double w = p.map_max_x - p.map_min_x, hgt = p.map_max_y - p.map_min_y;
    k.cx = (p.map_max_x + p.map_min_x) / 2; k.cy = (p.map_max_y + p.map_min_y) / 2;
    k.inv_hx = 1.0 / (w / 2); k.inv_hy = 1.0 / (hgt / 2);
    k.inv_M = 1.0 / std::sqrt(w * w + hgt * hgt);
    k.max_steer = p.max_steer;
    k.pos_thr = p.position_threshold; k.ori_thr = p.orientation_threshold;
    k.step_length = p.step_length;
    k.inv_step_length = 1.0 / p.step_length;
    for (int i = 0; i < 3; ++i) { k.rlo[i] = p.reset_lo[i]; k.rhi[i] = p.reset_hi[i]; }
    k.extra_steps = p.extra_steps; k.fixed_max = p.fixed_max_steps;
    k.term_mask = p.term_mask;
    k.npad = npad;
    k.stateless = p.stateless_reward;
    k.pool_m = 0;
    k.pool = nullptr;
    k.nt = nt_stores_for(npad);
#if TT_TABLE
    k.t = make_table();
#endif
    return k;
}

inline int grid_for(int n) { return (n + BLOCK - 1) / BLOCK; }

template <bool PER_ENV, bool INFO, bool RANDOM_POLICY>
void launch_step(tt_env *e, bool auto_reset, const float *action, float *action_out, float *obs, float *reward,
                 uint8_t *done, const Info &info, uint64_t policy_seed, hipStream_t s, const int *cursor = nullptr) {
    const dim3 g(grid_for(e->n)), b(BLOCK);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (e->profiling && e->ev_used < e->ev_start.size()) {
        t0 = e->ev_start[e->ev_used];
        t1 = e->ev_stop[e->ev_used];
        e->ev_used += 1;
    }
    // hipExtLaunchKernelGGL with null events is a plain launch; with events they time this dispatch alone
    if (auto_reset)
        hipExtLaunchKernelGGL((k_step<PER_ENV, INFO, true, RANDOM_POLICY>), g, b, 0, s, t0, t1, 0, e->kp, e->n, e->b,
                              action, action_out, obs, reward, done, info, e->seed, policy_seed, cursor);
    else
        hipExtLaunchKernelGGL((k_step<PER_ENV, INFO, false, RANDOM_POLICY>), g, b, 0, s, t0, t1, 0, e->kp, e->n, e->b,
                              action, action_out, obs, reward, done, info, e->seed, policy_seed, cursor);
}

template <bool RANDOM_POLICY>
int step_common(tt_env *env, const float *action, float *action_out, float *obs, float *reward, uint8_t *done,
                const tt_info *info, int auto_reset, uint64_t policy_seed, hipStream_t stream, const int *cursor = nullptr) {
    Info ki{nullptr, nullptr, nullptr};
    bool want_info = false;
    if (info) {
        ki.comp = info->comp; ki.violation = info->violation; ki.flags = info->flags;
        want_info = ki.comp || ki.violation || ki.flags;
    }
    const bool ar = auto_reset != 0;
    if (env->per_env) {
        if (want_info) launch_step<true, true, RANDOM_POLICY>(env, ar, action, action_out, obs, reward, done, ki, policy_seed, stream, cursor);
        else launch_step<true, false, RANDOM_POLICY>(env, ar, action, action_out, obs, reward, done, ki, policy_seed, stream, cursor);
    } else {
        if (want_info) launch_step<false, true, RANDOM_POLICY>(env, ar, action, action_out, obs, reward, done, ki, policy_seed, stream, cursor);
        else launch_step<false, false, RANDOM_POLICY>(env, ar, action, action_out, obs, reward, done, ki, policy_seed, stream, cursor);
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(env, TT_EHIP, "tt_env_step launch: %s", hipGetErrorString(err));
    return TT_OK;
}

}  // namespace

extern "C" {

int tt_version(void) { return TT_VERSION; }
#ifdef TT_STAMPS
int tt_debug_log_step(unsigned long long *out, int reset) {      // out: 1 + 2 * TT_LOG_CAP words (count, then begin / end pairs) or NULL
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_log_step), sizeof(TTLog)) != hipSuccess) return -3;
    if (reset) { const unsigned long long z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_log_step), &z, sizeof(z)) != hipSuccess) return -3; }
    return 0;
}
int tt_debug_step_blocks(unsigned long long *out2048) {
    return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_stepblk), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : -3;
}
#endif

const char *tt_last_error(const tt_env *env) { return env ? env->err : g_err; }

int tt_params_default(int variant, tt_params *out) {
    if (!out || (variant != 0 && variant != 1)) return fail(nullptr, TT_EINVAL, "tt_params_default: bad argument");
    std::memset(out, 0, sizeof(*out));
    out->L1 = variant ? 5.74 : 5.0;
    out->L2 = variant ? 10.192 : 7.0;
    out->hitch_offset = 0.0;
    out->v1x = -5.012;
    out->dt = 0.08;
    out->map_min_x = out->map_min_y = -40.0;
    out->map_max_x = out->map_max_y = 40.0;
    out->max_steer = 45.0 * kDeg;
    out->position_threshold = 0.5;
    out->orientation_threshold = 15.0 * kDeg;
    out->step_length = 0.40096;
    out->extra_steps = 75;
    out->fixed_max_steps = variant ? 300 : 0;
    out->term_mask = variant ? TT_TERM_SIMV1 : TT_TERM_SIMV2;
    out->variant = variant;
    out->stateless_reward = variant ? 1 : 0;
    out->goal[0] = 0.0; out->goal[1] = -30.0; out->goal[2] = 90.0 * kDeg;
    out->reset_lo[0] = -27.0; out->reset_hi[0] = 27.0;
    out->reset_lo[1] = 0.0; out->reset_hi[1] = 27.0;
    out->reset_lo[2] = 45.0 * kDeg; out->reset_hi[2] = 120.0 * kDeg;
    return TT_OK;
}

int tt_env_create(int n_envs, int device, const tt_params *params, tt_env **out) {
    if (!out) return fail(nullptr, TT_EINVAL, "tt_env_create: out is NULL");
    *out = nullptr;
    if (n_envs <= 0) return fail(nullptr, TT_EINVAL, "tt_env_create: n_envs must be positive (got %d)", n_envs);
    tt_params p;
    if (params) p = *params;
    else tt_params_default(0, &p);
    if (!(p.L1 > 0.0) || !(p.L2 > 0.0) || !(p.dt > 0.0) || !(p.step_length > 0.0) || !(p.map_max_x > p.map_min_x) ||
        !(p.map_max_y > p.map_min_y))
        return fail(nullptr, TT_EINVAL, "tt_env_create: non-physical parameters");
    // the in-step heading increments must stay small-angle (tt_sincos_small): dt*|v|*tan(max_steer)/L1 and dt*|v|/L2
    const double turn = std::fabs(p.dt * p.v1x) * std::fmax(std::tan(std::fabs(p.max_steer)) / p.L1, 1.0 / p.L2);
    if (!(turn <= 0.25) || !(std::fabs(p.max_steer) < 1.5))
        return fail(nullptr, TT_EINVAL, "tt_env_create: dt*|v|/L = %.3f rad per step exceeds the 0.25 rad the integrator's "
                                        "small-angle stage rotations are sized for", turn);
    {   // episode lengths must fit the 12-bit packed counters (steps | max_episode_steps)
        const double w = p.map_max_x - p.map_min_x, hgt = p.map_max_y - p.map_min_y;
        const double longest = p.fixed_max_steps > 0 ? (double)p.fixed_max_steps
                                                     : std::sqrt(w * w + hgt * hgt) * 1.5 / p.step_length + p.extra_steps;
        if (p.fixed_max_steps < 0 || p.extra_steps < 0 || !(longest <= (double)TT_MAX_EPISODE_STEPS))
            return fail(nullptr, TT_EINVAL, "tt_env_create: episodes of up to %.0f steps exceed the %d the packed counters hold",
                        longest, TT_MAX_EPISODE_STEPS);
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, TT_ENODEV, "tt_env_create: no HIP device visible");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= count) return fail(nullptr, TT_ENODEV, "tt_env_create: device %d of %d", device, count);
    tt_env *e = new (std::nothrow) tt_env;
    if (!e) return fail(nullptr, TT_ENOMEM, "tt_env_create: host allocation failed");
    e->n = n_envs;
    e->npad = (n_envs + TILE - 1) / TILE * TILE;
    e->device = device;
    e->params = p;
    e->kp = make_kparams(p, e->npad);
    const size_t npad = (size_t)e->npad;
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess) err = hipMalloc(&e->b.hot, sizeof(double) * H_ROWS * npad);
    if (err == hipSuccess) err = hipMalloc(&e->b.cold, sizeof(double) * C_ROWS * npad);
    if (err == hipSuccess) err = hipMalloc(&e->b.episodes, sizeof(uint32_t) * npad);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(k_init, dim3(grid_for(e->npad)), dim3(BLOCK), 0, nullptr, e->kp, e->b);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(nullptr);
    if (err != hipSuccess) {
        const int code = err == hipErrorOutOfMemory ? TT_ENOMEM : TT_EHIP;
        fail(nullptr, code, "tt_env_create: %s", hipGetErrorString(err));
        tt_env_destroy(e);
        return code;
    }
    *out = e;
    return TT_OK;
}

int tt_env_destroy(tt_env *env) {
    if (!env) return TT_OK;
    (void)hipSetDevice(env->device);
    if (env->b.hot) (void)hipFree(env->b.hot);
    if (env->b.cold) (void)hipFree(env->b.cold);
    if (env->b.episodes) (void)hipFree(env->b.episodes);
    for (hipEvent_t ev : env->ev_start) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : env->ev_stop) (void)hipEventDestroy(ev);
    delete env;
    return TT_OK;
}

int tt_env_num_envs(const tt_env *env) { return env ? env->n : TT_EINVAL; }

int tt_env_reset(tt_env *env, const uint8_t *mask, uint64_t seed, float *obs_out, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_reset: NULL handle");
    TT_HIP(env, hipSetDevice(env->device));
    env->seed = seed;
    hipLaunchKernelGGL(k_reset, dim3(grid_for(env->n)), dim3(BLOCK), 0, stream, env->kp, env->n, env->b, mask, obs_out,
                       env->seed);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_set_reset_pool(tt_env *env, const double *pool, int m) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_reset_pool: NULL handle");
    if (m < 0) return fail(env, TT_EINVAL, "tt_env_set_reset_pool: m=%d", m);
    env->kp.pool_m = pool ? m : 0;
    env->kp.pool = env->kp.pool_m > 0 ? pool : nullptr;
    return TT_OK;
}

int tt_env_set_pose(tt_env *env, const int32_t *idx, int k, const double *start, const double *goal, const double *L2,
                    float *obs_out, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_pose: NULL handle");
    if (k < 0 || k > env->n || (k > 0 && !start))
        return fail(env, TT_EINVAL, "tt_env_set_pose: k=%d outside [0,%d] or start NULL", k, env->n);
    if (k == 0) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    if (goal || L2) env->per_env = true;
    hipLaunchKernelGGL(k_set_pose, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->kp, env->n, env->b, idx, k, start, goal,
                       L2, obs_out);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_set_attrs(tt_env *env, const int32_t *idx, int k, const double *start, const double *goal, const double *L2,
                     tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_attrs: NULL handle");
    if (k < 0 || k > env->n) return fail(env, TT_EINVAL, "tt_env_set_attrs: k=%d outside [0,%d]", k, env->n);
    if (k == 0 || (!start && !goal && !L2)) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    if (goal || L2) env->per_env = true;
    hipLaunchKernelGGL(k_set_attrs, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->kp, env->n, env->b, idx, k, start, goal,
                       L2);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_set_state(tt_env *env, const int32_t *idx, int k, const double *state, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_state: NULL handle");
    if (k < 0 || k > env->n || (k > 0 && !state))
        return fail(env, TT_EINVAL, "tt_env_set_state: k=%d outside [0,%d] or state NULL", k, env->n);
    if (k == 0) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    hipLaunchKernelGGL(k_set_state, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->n, env->b, idx, k, state);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_get_state(tt_env *env, double *state_out, tt_stream_t stream) {
    if (!env || !state_out) return fail(env, TT_EINVAL, "tt_env_get_state: NULL argument");
    TT_HIP(env, hipSetDevice(env->device));
    hipLaunchKernelGGL(k_get_state, dim3(grid_for(env->n)), dim3(BLOCK), 0, stream, env->n, env->b, state_out);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_set_max_steps(tt_env *env, const int32_t *idx, int k, const int32_t *max_steps, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_max_steps: NULL handle");
    if (k < 0 || k > env->n || (k > 0 && !max_steps))
        return fail(env, TT_EINVAL, "tt_env_set_max_steps: k=%d outside [0,%d] or max_steps NULL", k, env->n);
    if (k == 0) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    {   // the packed counters hold 12 bits: refuse what they cannot represent instead of clamping (not a hot call)
        std::vector<int32_t> host((size_t)k);
        TT_HIP(env, hipMemcpyAsync(host.data(), max_steps, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost, stream));
        TT_HIP(env, hipStreamSynchronize(stream));
        for (int j = 0; j < k; ++j)
            if (host[(size_t)j] < 0 || host[(size_t)j] > TT_MAX_EPISODE_STEPS)
                return fail(env, TT_EINVAL, "tt_env_set_max_steps: max_steps[%d] = %d outside [0, %d]", j, host[(size_t)j],
                            TT_MAX_EPISODE_STEPS);
    }
    hipLaunchKernelGGL(k_set_max_steps, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->n, env->b, idx, k, max_steps);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_set_steps(tt_env *env, const int32_t *idx, int k, const int32_t *steps, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_steps: NULL handle");
    if (k < 0 || k > env->n || (k > 0 && !steps))
        return fail(env, TT_EINVAL, "tt_env_set_steps: k=%d outside [0,%d] or steps NULL", k, env->n);
    if (k == 0) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    {   // 12-bit packed counter, as for tt_env_set_max_steps: refuse instead of clamping (a setter, not a hot call)
        std::vector<int32_t> host((size_t)k);
        TT_HIP(env, hipMemcpyAsync(host.data(), steps, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost, stream));
        TT_HIP(env, hipStreamSynchronize(stream));
        for (int j = 0; j < k; ++j)
            if (host[(size_t)j] < 0 || host[(size_t)j] > TT_MAX_EPISODE_STEPS)
                return fail(env, TT_EINVAL, "tt_env_set_steps: steps[%d] = %d outside [0, %d]", j, host[(size_t)j],
                            TT_MAX_EPISODE_STEPS);
    }
    hipLaunchKernelGGL(k_set_steps, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->n, env->b, idx, k, steps);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_get_episode(tt_env *env, int32_t *steps, int32_t *max_steps, double *start, double *goal, double *L2,
                       tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_get_episode: NULL handle");
    if (!steps && !max_steps && !start && !goal && !L2) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    hipLaunchKernelGGL(k_get_episode, dim3(grid_for(env->n)), dim3(BLOCK), 0, stream, env->kp, env->n, env->b, steps,
                       max_steps, start, goal, L2);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_observe(tt_env *env, const float *steering, float *obs_out, tt_stream_t stream) {
    if (!env || !obs_out) return fail(env, TT_EINVAL, "tt_env_observe: NULL argument");
    TT_HIP(env, hipSetDevice(env->device));
    const dim3 g(grid_for(env->n)), b(BLOCK);
    if (env->per_env)
        hipLaunchKernelGGL(k_observe<true>, g, b, 0, stream, env->kp, env->n, env->b, steering, obs_out);
    else
        hipLaunchKernelGGL(k_observe<false>, g, b, 0, stream, env->kp, env->n, env->b, steering, obs_out);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_set_step_counter(tt_env *env, int64_t *counter) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_step_counter: NULL handle");
    env->b.counter = reinterpret_cast<long long *>(counter);
    return TT_OK;
}

int tt_env_step(tt_env *env, const float *action, float *obs, float *reward, uint8_t *done, const tt_info *info,
                int auto_reset, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_step: NULL handle");
    if (!action || !obs || !reward || !done)
        return fail(env, TT_EINVAL, "tt_env_step: action, obs, reward and done are required");
    TT_HIP(env, hipSetDevice(env->device));
    return step_common<false>(env, action, nullptr, obs, reward, done, info, auto_reset, 0, stream);
}

int tt_env_step_ring(tt_env *env, const float *action, const tt_ring_view *ring, int auto_reset, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_step_ring: NULL handle");
    if (!action || !ring || !ring->cursor || !ring->obs || !ring->rew || !ring->done || ring->n_envs != env->n)
        return fail(env, TT_EINVAL, "tt_env_step_ring: action and a ring view of this handle's %d envs are required", env->n);
    TT_HIP(env, hipSetDevice(env->device));
    return step_common<false>(env, action, nullptr, ring->obs, ring->rew, ring->done, nullptr, auto_reset, 0, stream, ring->cursor);
}

int tt_env_step_random(tt_env *env, uint64_t policy_seed, float *action_out, float *obs, float *reward, uint8_t *done,
                       const tt_info *info, int auto_reset, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_step_random: NULL handle");
    if (!obs || !reward || !done) return fail(env, TT_EINVAL, "tt_env_step_random: obs, reward and done are required");
    TT_HIP(env, hipSetDevice(env->device));
    return step_common<true>(env, nullptr, action_out, obs, reward, done, info, auto_reset, policy_seed, stream);
}

size_t tt_env_state_bytes(const tt_env *env) {
    if (!env) return 0;
    const size_t npad = (size_t)env->npad;
    return sizeof(double) * (H_ROWS + C_ROWS) * npad + sizeof(uint32_t) * npad;
}

int tt_env_export(tt_env *env, void *blob, uint64_t meta[4], tt_stream_t stream) {
    if (!env || !blob || !meta) return fail(env, TT_EINVAL, "tt_env_export: NULL argument");
    TT_HIP(env, hipSetDevice(env->device));
    const size_t npad = (size_t)env->npad, hb = sizeof(double) * H_ROWS * npad, cb = sizeof(double) * C_ROWS * npad;
    char *dst = static_cast<char *>(blob);
    TT_HIP(env, hipMemcpyAsync(dst, env->b.hot, hb, hipMemcpyDeviceToDevice, stream));
    TT_HIP(env, hipMemcpyAsync(dst + hb, env->b.cold, cb, hipMemcpyDeviceToDevice, stream));
    TT_HIP(env, hipMemcpyAsync(dst + hb + cb, env->b.episodes, sizeof(uint32_t) * npad, hipMemcpyDeviceToDevice, stream));
    meta[0] = env->per_env ? 1 : 0; meta[1] = env->seed; meta[2] = (uint64_t)env->n; meta[3] = TT_VERSION;
    return TT_OK;
}

int tt_env_import(tt_env *env, const void *blob, const uint64_t meta[4], tt_stream_t stream) {
    if (!env || !blob || !meta) return fail(env, TT_EINVAL, "tt_env_import: NULL argument");
    if (meta[2] != (uint64_t)env->n || meta[3] != TT_VERSION)
        return fail(env, TT_EINVAL, "tt_env_import: blob is for %llu envs (version %llu), handle has %d (version %d)",
                    (unsigned long long)meta[2], (unsigned long long)meta[3], env->n, TT_VERSION);
    TT_HIP(env, hipSetDevice(env->device));
    const size_t npad = (size_t)env->npad, hb = sizeof(double) * H_ROWS * npad, cb = sizeof(double) * C_ROWS * npad;
    const char *src = static_cast<const char *>(blob);
    TT_HIP(env, hipMemcpyAsync(env->b.hot, src, hb, hipMemcpyDeviceToDevice, stream));
    TT_HIP(env, hipMemcpyAsync(env->b.cold, src + hb, cb, hipMemcpyDeviceToDevice, stream));
    TT_HIP(env, hipMemcpyAsync(env->b.episodes, src + hb + cb, sizeof(uint32_t) * npad, hipMemcpyDeviceToDevice, stream));
    env->per_env = meta[0] != 0;
    env->seed = meta[1];
    return TT_OK;
}

int tt_env_rollout_random(tt_env *env, int k_steps, uint64_t policy_seed, float *obs_out, float *reward_sum,
                          int32_t *episodes_done, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_rollout_random: NULL handle");
    if (k_steps < 0) return fail(env, TT_EINVAL, "tt_env_rollout_random: k_steps=%d", k_steps);
    if (k_steps == 0) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    const dim3 g(grid_for(env->n)), b(BLOCK);
    if (env->per_env)
        hipLaunchKernelGGL(k_rollout<true>, g, b, 0, stream, env->kp, env->n, env->b, k_steps, obs_out, reward_sum,
                           episodes_done, env->seed, policy_seed);
    else
        hipLaunchKernelGGL(k_rollout<false>, g, b, 0, stream, env->kp, env->n, env->b, k_steps, obs_out, reward_sum,
                           episodes_done, env->seed, policy_seed);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_profile(tt_env *env, int max_launches) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_profile: NULL handle");
    TT_HIP(env, hipSetDevice(env->device));
    env->profiling = max_launches > 0;
    env->ev_used = 0;
    env->prof_ms = 0.0;
    env->prof_launches = 0;
    while ((long)env->ev_start.size() < (long)max_launches) {
        hipEvent_t a, b;
        TT_HIP(env, hipEventCreate(&a));
        TT_HIP(env, hipEventCreate(&b));
        env->ev_start.push_back(a);
        env->ev_stop.push_back(b);
    }
    return TT_OK;
}

int tt_env_profile_read(tt_env *env, double *total_ms, int64_t *launches) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_profile_read: NULL handle");
    TT_HIP(env, hipSetDevice(env->device));
    for (size_t i = 0; i < env->ev_used; ++i) {
        TT_HIP(env, hipEventSynchronize(env->ev_stop[i]));
        float ms = 0.f;
        TT_HIP(env, hipEventElapsedTime(&ms, env->ev_start[i], env->ev_stop[i]));
        env->prof_ms += ms;
        env->prof_launches += 1;
    }
    env->ev_used = 0;
    if (total_ms) *total_ms = env->prof_ms;
    if (launches) *launches = env->prof_launches;
    return TT_OK;
}

int tt_random_actions(int n, uint64_t seed, uint64_t step, float *out, tt_stream_t stream) {
    if (n < 0 || (n > 0 && !out)) return fail(nullptr, TT_EINVAL, "tt_random_actions: bad argument");
    if (n == 0) return TT_OK;
    hipLaunchKernelGGL(k_random_actions, dim3(grid_for(n)), dim3(BLOCK), 0, stream, n, seed, step, out);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(nullptr, TT_EHIP, "tt_random_actions: %s", hipGetErrorString(err));
    return TT_OK;
}

}  // extern "C"
