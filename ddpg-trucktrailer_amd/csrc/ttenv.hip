// ttenv.hip -- libttenv.so: batched truck-trailer backing environment for MI355X (gfx950).
//
// One thread per env, SoA f64 state in HBM, one fused kernel per vector step:
//   clip -> one Dormand-Prince step of the 6-state kinematic ODE (f64) -> 23-dim observation
//   -> reward_functionv1 reward with its carry -> termination flags -> (optional) in-kernel reset.
// The observation tile of a workgroup is staged through LDS so that each wave stores whole
// 16-byte vectors of the row-major [N,23] f32 matrix instead of 64 words 92 bytes apart.
//
// What is restated from the reference (paths relative to pain7576/ddpg-trucktrailer):
//   kinematic ODE            truck_trailer_sim/simv2.py:269-303
//   observation              truck_trailer_sim/simv2.py:103-181
//   reset / pose override    truck_trailer_sim/simv2.py:459-498, 263-267; DDPG/test.py:96-115
//   step, flags, done        truck_trailer_sim/simv2.py:499-545, 305-345
//   reward + carry           truck_trailer_sim/reward_functionv1.py:6-109, 144-506
//   integrator               scipy RK45 tableau (scipy/integrate/_ivp/rk.py), ONE step of h = dt
//
// Arithmetic shortcuts taken here (the oracle in oracle/tt_oracle.c takes none, so the parity
// tests check them): the hitch-angle sin/cos come from the angle-difference identities on the
// sin/cos of the two headings; obs[19..22] come from the same identities and from
// (dx, dy)/distance instead of atan2 followed by sin/cos; cos(wrap(x)) = cos(x).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "ttenv.h"

namespace {

constexpr int OBS = TT_OBS_DIM;
constexpr int BLOCK = 256;
constexpr double kPi = 3.14159265358979323846;
constexpr double kDeg = kPi / 180.0;

// rows of the [R_COUNT, N] f64 SoA block
enum Row : int {
    R_PSI1 = 0, R_PSI2, R_X1, R_Y1, R_X2, R_Y2,       // kinematic state
    R_D3, R_D2, R_D1, R_PREV, R_CUM, R_CLOSEST,       // reward carry: distance window, backward sum, closest
    R_DINIT,                                          // |goal - start| (without the 1e-6 of reward_functionv1.py:35)
    R_SX, R_SY, R_SYAW,                               // start pose (read back only)
    R_GX, R_GY, R_GYAW, R_SG, R_CG, R_L2,             // per-env goal (+ sin/cos of its yaw) and trailer length
    R_COUNT
};

// packed per-env counters: steps [0,12) | max_episode_steps [12,24) | stages_achieved [24,27)
__host__ __device__ inline uint32_t pk_steps(uint32_t p) { return p & 0xFFFu; }
__host__ __device__ inline uint32_t pk_max(uint32_t p) { return (p >> 12) & 0xFFFu; }
__host__ __device__ inline uint32_t pk_stages(uint32_t p) { return (p >> 24) & 0x7u; }
__host__ __device__ inline uint32_t pk_make(uint32_t steps, uint32_t maxs, uint32_t stages) {
    return (steps > 0xFFFu ? 0xFFFu : steps) | ((maxs > 0xFFFu ? 0xFFFu : maxs) << 12) | ((stages & 7u) << 24);
}

struct KParams {
    double v, v_over_L1, ho, h;
    double L2;
    double gx, gy, gyaw, sg, cg;
    double minx, maxx, miny, maxy;
    double cx, cy, inv_hx, inv_hy, inv_M;
    double max_steer, pos_thr, ori_thr, step_length;
    double rlo[3], rhi[3];
    int extra_steps, fixed_max;
    unsigned term_mask;
};

struct Info {
    double *comp;
    uint8_t *violation;
    uint8_t *flags;
};

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
struct Goal {
    double gx, gy, sg, cg;
};

// 23-dim observation in f64 from the state and the sin/cos the caller already has (simv2.py:103-181).
// Returns the trailer-goal distance.
__device__ inline double observe(const KParams &P, const Goal &g, double x1, double y1, double x2, double y2,
                                 double s1, double c1, double s2, double c2, double sd, double cd, float *of) {
    const double dx = g.gx - x2, dy = g.gy - y2;
    const double cur = sqrt(dx * dx + dy * dy);
    const double dxl = dx * c2 + dy * s2;
    const double dyl = -dx * s2 + dy * c2;
    double sh, ch;  // sin/cos of atan2(dy,dx) - (psi2 + pi)
    if (cur > 0.0) {
        const double inv = 1.0 / cur;
        sh = -dyl * inv;
        ch = -dxl * inv;
    } else {  // atan2(0, 0) = 0
        sh = s2;
        ch = -c2;
    }
    of[0] = (float)((x1 - P.cx) * P.inv_hx);
    of[1] = (float)((y1 - P.cy) * P.inv_hy);
    of[2] = (float)s1;
    of[3] = (float)c1;
    of[4] = (float)((x2 - P.cx) * P.inv_hx);
    of[5] = (float)((y2 - P.cy) * P.inv_hy);
    of[6] = (float)s2;
    of[7] = (float)c2;
    of[8] = (float)(s1 * c2 - c1 * s2);
    of[9] = (float)(c1 * c2 + s1 * s2);
    of[10] = (float)sd;
    of[11] = (float)cd;
    of[12] = (float)((g.gx - P.cx) * P.inv_hx);
    of[13] = (float)((g.gy - P.cy) * P.inv_hy);
    of[14] = (float)g.sg;
    of[15] = (float)g.cg;
    of[16] = (float)fmin(fmax(cur * P.inv_M, 0.0), 1.0);
    of[17] = (float)fmin(fmax(dxl * P.inv_M, -1.0), 1.0);
    of[18] = (float)fmin(fmax(dyl * P.inv_M, -1.0), 1.0);
    of[19] = (float)(g.sg * c2 - g.cg * s2);
    of[20] = (float)(g.cg * c2 + g.sg * s2);
    of[21] = (float)sh;
    of[22] = (float)ch;
    return cur;
}

// Store a workgroup's [nv,23] f32 observation tile, staged through LDS so the global stores are
// contiguous 16-byte vectors.  Every thread of the block must call this (barriers inside).
__device__ inline void store_obs_tile(float *tile, const float *of, bool valid, float *obs, int block_first, int nv) {
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
        for (int q = tid; q < nvec; q += BLOCK) d4[q] = t4[q];
        for (int q = (nvec << 2) + tid; q < total; q += BLOCK) dst[q] = tile[q];
    } else {
        for (int q = tid; q < total; q += BLOCK) dst[q] = tile[q];
    }
    __syncthreads();
}

// Place one env at a start pose (simv2.py:481-496 / DDPG/test.py:96-115).  Returns the packed counters.
__device__ inline uint32_t place(const KParams &P, double *f, int n, int i, double sx, double sy, double syaw,
                                 const Goal &g, double gyaw, double L2, float *of) {
    double ss, cs;
    sincos(syaw, &ss, &cs);
    // state is stored as float32 by the reference (simv2.py:489)
    const double psi = (double)(float)syaw;
    const double x1 = (double)(float)(sx + L2 * cs), y1 = (double)(float)(sy + L2 * ss);
    const double x2 = (double)(float)sx, y2 = (double)(float)sy;
    f[R_PSI1 * (size_t)n + i] = psi;
    f[R_PSI2 * (size_t)n + i] = psi;
    f[R_X1 * (size_t)n + i] = x1;
    f[R_Y1 * (size_t)n + i] = y1;
    f[R_X2 * (size_t)n + i] = x2;
    f[R_Y2 * (size_t)n + i] = y2;
    const double ddx = g.gx - sx, ddy = g.gy - sy;
    const double dinit = sqrt(ddx * ddx + ddy * ddy);
    f[R_DINIT * (size_t)n + i] = dinit;
    f[R_SX * (size_t)n + i] = sx;
    f[R_SY * (size_t)n + i] = sy;
    f[R_SYAW * (size_t)n + i] = syaw;
    f[R_GX * (size_t)n + i] = g.gx;
    f[R_GY * (size_t)n + i] = g.gy;
    f[R_GYAW * (size_t)n + i] = gyaw;
    f[R_SG * (size_t)n + i] = g.sg;
    f[R_CG * (size_t)n + i] = g.cg;
    f[R_L2 * (size_t)n + i] = L2;
    const int maxs = P.fixed_max > 0 ? P.fixed_max : (int)(dinit / P.step_length) + P.extra_steps;
    if (of) {
        double sp, cp;
        sincos(psi, &sp, &cp);
        observe(P, g, x1, y1, x2, y2, sp, cp, sp, cp, 0.0, 1.0, of);
    }
    return pk_make(0u, (uint32_t)maxs, 0u);
}

__device__ inline void random_pose(const KParams &P, uint64_t seed, uint32_t env, uint64_t nonce, double &sx, double &sy,
                                   double &syaw) {
    uint32_t r[4];
    philox4x32(env, (uint32_t)nonce, (uint32_t)(nonce >> 32), 0x7452u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    sx = P.rlo[0] + (P.rhi[0] - P.rlo[0]) * u01(r[0]);  // draw order x, y, yaw (simv2.py:331-333)
    sy = P.rlo[1] + (P.rhi[1] - P.rlo[1]) * u01(r[1]);
    syaw = P.rlo[2] + (P.rhi[2] - P.rlo[2]) * u01(r[2]);
}

// ------------------------------------------------------------------------------------------
// the vector step
template <bool PER_ENV, bool INFO, bool AUTO_RESET>
__global__ __launch_bounds__(BLOCK) void k_step(const KParams P, const int n, double *__restrict__ f,
                                                float *__restrict__ prev_steer, uint32_t *__restrict__ packed,
                                                const float *__restrict__ action, float *__restrict__ obs,
                                                float *__restrict__ reward, uint8_t *__restrict__ done, const Info info,
                                                const uint64_t seed, const uint64_t nonce) {
    __shared__ __attribute__((aligned(16))) float tile[BLOCK * OBS];
    const int block_first = blockIdx.x * BLOCK;
    const int i = block_first + threadIdx.x;
    const bool valid = i < n;
    const int nv = min(BLOCK, n - block_first);
    float of[OBS];

    if (valid) {
        const size_t N = (size_t)n;
        double psi1 = f[R_PSI1 * N + i], psi2 = f[R_PSI2 * N + i];
        double x1 = f[R_X1 * N + i], y1 = f[R_Y1 * N + i], x2 = f[R_X2 * N + i], y2 = f[R_Y2 * N + i];
        double d3 = f[R_D3 * N + i], d2 = f[R_D2 * N + i], d1 = f[R_D1 * N + i], prev = f[R_PREV * N + i];
        double cum = f[R_CUM * N + i], closest = f[R_CLOSEST * N + i];
        const double dinit = f[R_DINIT * N + i];
        float psteer = prev_steer[i];
        const uint32_t pk = packed[i];
        Goal g;
        double L2;
        if (PER_ENV) {
            g.gx = f[R_GX * N + i]; g.gy = f[R_GY * N + i]; g.sg = f[R_SG * N + i]; g.cg = f[R_CG * N + i];
            L2 = f[R_L2 * N + i];
        } else {
            g.gx = P.gx; g.gy = P.gy; g.sg = P.sg; g.cg = P.cg;
            L2 = P.L2;
        }

        // ---- simv2.py:504-505: clip in f64 against np.radians(45)
        const double delta = fmin(fmax((double)action[i], -P.max_steer), P.max_steer);
        double sd, cd;
        sincos(delta, &sd, &cd);
        const double w1 = P.v_over_L1 * (sd / cd);  // truck yaw rate, constant over the step

        // ---- one Dormand-Prince step (scipy RK45 tableau).  Only the two headings feed back into
        // the right-hand side, so the positions are accumulated straight into their B-weighted sums.
        const double h = P.h, v = P.v, vL2 = P.v / L2, hoL2 = P.ho / L2, how1 = P.ho * w1;
        constexpr double C[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
        constexpr double A[6][5] = {{0, 0, 0, 0, 0},
                                    {1.0 / 5, 0, 0, 0, 0},
                                    {3.0 / 40, 9.0 / 40, 0, 0, 0},
                                    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                                    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                                    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
        constexpr double B[6] = {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
        double k2[6];
        double ax1 = 0.0, ay1 = 0.0, ax2 = 0.0, ay2 = 0.0, apsi2 = 0.0;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < s; ++j) acc += A[s][j] * k2[j];
            const double p1 = psi1 + (h * C[s]) * w1;
            const double p2 = psi2 + h * acc;
            double s1, c1, s2, c2;
            sincos(p1, &s1, &c1);
            sincos(p2, &s2, &c2);
            const double sth = s1 * c2 - c1 * s2, cth = c1 * c2 + s1 * s2;
            const double v2 = v * cth + how1 * sth;
            k2[s] = vL2 * sth - hoL2 * w1 * cth;
            if (B[s] != 0.0) {
                apsi2 += B[s] * k2[s];
                ax1 += B[s] * (v * c1);
                ay1 += B[s] * (v * s1);
                ax2 += B[s] * (v2 * c2);
                ay2 += B[s] * (v2 * s2);
            }
        }
        psi1 += h * w1;
        psi2 += h * apsi2;
        x1 += h * ax1;
        y1 += h * ay1;
        x2 += h * ax2;
        y2 += h * ay2;

        // ---- observation (simv2.py:519), cast to f32 like np.array(..., dtype=float32)
        double s1, c1, s2, c2;
        sincos(psi1, &s1, &c1);
        sincos(psi2, &s2, &c2);
        const double cur = observe(P, g, x1, y1, x2, y2, s1, c1, s2, c2, sd, cd, of);

        // ---- reward (reward_functionv1.py:442-506); episode_steps is incremented first (simv2.py:523)
        const uint32_t steps = pk_steps(pk) + 1u;
        uint32_t stages = pk_stages(pk);
        const double init = dinit + 1e-6;
        const int rmax = (int)(init / P.step_length) + P.extra_steps;
        const float steer_now = atan2f(of[10], of[11]);
        if (pk_steps(pk) == 0u) {  // first step of the episode: reward_state is None (:40-76)
            d3 = d2 = d1 = prev = cur;
            cum = 0.0;
            closest = cur;
            psteer = steer_now;
            stages = 0u;
        } else if (cur < closest) {
            closest = cur;
        }
        // dynamic weights (:189-238)
        const double jp = fmin(fmax((init - cur) / init, 0.0), 1.0);
        const double w_orient = (tanh(7.0 * (jp - 0.3)) + 1.0) * 0.5;
        const double w_head = 1.0 - w_orient;
        // progress (:144-187)
        const double inst = prev - cur;
        const double th = tanh(inst);
        const double prog_net = tanh((d3 - cur) * 0.5) * 0.5;
        const bool mono = (d1 >= prev) && (prev >= cur);
        const double progress = (inst > 0.0 ? th : th * 0.5) + prog_net + (mono ? 0.2 : 0.0);
        // heading (:285-309): cos(atan2(dy,dx) - (atan2f(o6,o7) + pi)), the orientation read back in f32
        double sco, cco;
        sincos((double)atan2f(of[6], of[7]), &sco, &cco);
        double heading;
        {
            const double dx = g.gx - x2, dy = g.gy - y2;
            heading = cur > 0.0 ? -(dx * cco + dy * sco) / cur : -cco;
        }
        // orientation (:311-324): f32 * 15.0 stays f32 in numpy
        const float orient15 = of[20] * 15.0f;
        // staged bonuses (:338-367)
        const double ori_err = (double)fabsf(atan2f(of[19], of[20]));
        double staged = 0.0;
        if (cur <= 5.0) { staged += 10.0; stages |= 1u; }
        if (cur <= 2.0 && ori_err <= 45.0 * kDeg && !(stages & 2u)) { staged += 25.0; stages |= 2u; }
        const bool at_goal = cur <= P.pos_thr && ori_err <= P.ori_thr;
        if (at_goal && !(stages & 4u)) { staged += 100.0; stages |= 4u; }
        // safety (:369-421)
        double safety = 0.0;
        uint32_t viol = TT_V_NONE;
        const double hitch = fabs(psi1 - psi2);
        if (hitch > 85.0 * kDeg) { safety += -500.0; viol = TT_V_JACKKNIFE; }
        else if (hitch > 70.0 * kDeg) { safety += -50.0; viol = TT_V_JACKKNIFE_WARNING; }
        const double lox = fmin(x1, x2), hix = fmax(x1, x2), loy = fmin(y1, y2), hiy = fmax(y1, y2);
        const bool outside = lox < P.minx || hix > P.maxx || loy < P.miny || hiy > P.maxy;
        if (lox < P.minx - 2.0 || hix > P.maxx + 2.0 || loy < P.miny - 2.0 || hiy > P.maxy + 2.0) {
            safety += -500.0; viol = TT_V_MAJOR_BOUNDARY;
        } else if (outside) {
            safety += -50.0; viol = TT_V_MINOR_BOUNDARY;
        }
        const bool passed = g.gy > y2;
        if (passed) { safety += -500.0; viol = TT_V_PAST_THE_GOAL; }
        if ((int)steps >= rmax) { safety += -500.0; viol = TT_V_MAX_STEP; }
        const bool excessive = cur > closest + 6.0;  // :120-124
        if (excessive) { safety += -500.0; viol = TT_V_EXCESSIVE_BACKWARD; }
        // exploration (:423-439)
        const double explore = (double)steps < rmax * 0.5 ? 4.0 : ((double)steps < rmax * 0.8 ? 2.0 : 0.0);
        // backward-movement budget (:240-283)
        cum += fmax(0.0, cur - prev);
        const double budget = 5.0 * fmin(1.0, (double)steps / 50.0);
        const double excess = fmax(0.0, cum - budget);
        const double back = excess > 0.0 ? -(excess * sqrt(excess)) * 0.5 : 0.0;
        // smoothness against the episode's FIRST steering (:326-335; previous_steering is never refreshed)
        const double smooth = (double)fabsf(steer_now - psteer) / (90.0 * kDeg);
        const double final_bonus = at_goal ? 200.0 : 0.0;

        const double c_prog = progress * 15.0, c_head = heading * 15.0 * w_head;
        const double c_orient = (double)orient15 * w_orient, c_smooth = smooth * -25.0;
        const double total = 0.0 + c_prog + c_head + c_orient + staged + safety + explore + back + c_smooth + final_bonus;

        // ---- flags (simv2.py:528-541)
        uint32_t fl = 0u;
        if (hitch > 90.0 * kDeg) fl |= TT_F_JACKKNIFE;
        if (outside) fl |= TT_F_OUT_OF_MAP;
        if (steps >= pk_max(pk)) fl |= TT_F_MAX_STEPS;
        if (at_goal) fl |= TT_F_GOAL_REACHED | TT_F_SUCCESS;
        if (passed) fl |= TT_F_GOAL_PASSED;
        if (excessive) fl |= TT_F_EXCESSIVE_BACK;
        const bool is_done = (fl & P.term_mask) != 0u;

        reward[i] = (float)total;
        done[i] = is_done ? 1 : 0;
        if (INFO) {
            if (info.comp) {
                double *c = info.comp + i;
                c[TT_I_TOTAL * N] = total; c[TT_I_PROGRESS * N] = c_prog; c[TT_I_HEADING * N] = c_head;
                c[TT_I_ORIENT * N] = c_orient; c[TT_I_STAGED * N] = staged; c[TT_I_SAFETY * N] = safety;
                c[TT_I_EXPLORE * N] = explore; c[TT_I_FINAL * N] = final_bonus; c[TT_I_BACKWARD * N] = back;
                c[TT_I_SMOOTH * N] = c_smooth; c[TT_I_CUMBACK * N] = cum; c[TT_I_BUDGET * N] = budget;
            }
            if (info.violation) info.violation[i] = (uint8_t)viol;
            if (info.flags) info.flags[i] = (uint8_t)fl;
        }

        if (AUTO_RESET && is_done) {
            double sx, sy, syaw;
            random_pose(P, seed, (uint32_t)i, nonce, sx, sy, syaw);
            Goal g0{P.gx, P.gy, P.sg, P.cg};
            packed[i] = place(P, f, n, i, sx, sy, syaw, g0, P.gyaw, L2, of);
        } else {
            f[R_PSI1 * N + i] = psi1; f[R_PSI2 * N + i] = psi2;
            f[R_X1 * N + i] = x1; f[R_Y1 * N + i] = y1; f[R_X2 * N + i] = x2; f[R_Y2 * N + i] = y2;
            f[R_D3 * N + i] = d2; f[R_D2 * N + i] = d1; f[R_D1 * N + i] = prev; f[R_PREV * N + i] = cur;
            f[R_CUM * N + i] = cum; f[R_CLOSEST * N + i] = closest;
            prev_steer[i] = psteer;
            packed[i] = pk_make(steps, pk_max(pk), stages);
        }
    }
    store_obs_tile(tile, of, valid, obs, block_first, nv);
}

// ------------------------------------------------------------------------------------------
// reset / pose / state kernels (not hot)
__global__ __launch_bounds__(BLOCK) void k_reset(const KParams P, const int n, double *f, uint32_t *packed,
                                                 const uint8_t *mask, float *obs, const uint64_t seed,
                                                 const uint64_t nonce) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (mask && !mask[i])) return;
    double sx, sy, syaw;
    random_pose(P, seed, (uint32_t)i, nonce, sx, sy, syaw);
    float of[OBS];
    Goal g{P.gx, P.gy, P.sg, P.cg};
    packed[i] = place(P, f, n, i, sx, sy, syaw, g, P.gyaw, f[R_L2 * (size_t)n + i], obs ? of : nullptr);
    if (obs)
        for (int j = 0; j < OBS; ++j) obs[(size_t)i * OBS + j] = of[j];
}

__global__ __launch_bounds__(BLOCK) void k_init(const KParams P, const int n, double *f, float *prev_steer,
                                                uint32_t *packed) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    for (int r = 0; r < R_COUNT; ++r) f[r * (size_t)n + i] = 0.0;
    f[R_L2 * (size_t)n + i] = P.L2;
    Goal g{P.gx, P.gy, P.sg, P.cg};
    packed[i] = place(P, f, n, i, P.gx, P.gy + 30.0, kPi / 2, g, P.gyaw, P.L2, nullptr);
    prev_steer[i] = 0.0f;
}

__global__ __launch_bounds__(BLOCK) void k_set_pose(const KParams P, const int n, double *f, uint32_t *packed,
                                                    const int32_t *idx, const int k, const double *start,
                                                    const double *goal, const double *L2, float *obs) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    const size_t N = (size_t)n;
    Goal g;
    double gyaw;
    if (goal) {
        g.gx = goal[3 * j]; g.gy = goal[3 * j + 1]; gyaw = goal[3 * j + 2];
        sincos(gyaw, &g.sg, &g.cg);
    } else {
        g.gx = f[R_GX * N + i]; g.gy = f[R_GY * N + i]; gyaw = f[R_GYAW * N + i];
        g.sg = f[R_SG * N + i]; g.cg = f[R_CG * N + i];
    }
    const double l2 = L2 ? L2[j] : f[R_L2 * N + i];
    float of[OBS];
    packed[i] = place(P, f, n, i, start[3 * j], start[3 * j + 1], start[3 * j + 2], g, gyaw, l2, obs ? of : nullptr);
    if (obs)
        for (int q = 0; q < OBS; ++q) obs[(size_t)i * OBS + q] = of[q];
}

__global__ __launch_bounds__(BLOCK) void k_set_attrs(const int n, double *f, const int32_t *idx, const int k,
                                                     const double *start, const double *goal, const double *L2) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    const size_t N = (size_t)n;
    if (start) {
        f[R_SX * N + i] = start[3 * j]; f[R_SY * N + i] = start[3 * j + 1]; f[R_SYAW * N + i] = start[3 * j + 2];
    }
    if (goal) {
        double sg, cg;
        sincos(goal[3 * j + 2], &sg, &cg);
        f[R_GX * N + i] = goal[3 * j]; f[R_GY * N + i] = goal[3 * j + 1]; f[R_GYAW * N + i] = goal[3 * j + 2];
        f[R_SG * N + i] = sg; f[R_CG * N + i] = cg;
    }
    if (L2) f[R_L2 * N + i] = L2[j];
    const double ddx = f[R_GX * N + i] - f[R_SX * N + i], ddy = f[R_GY * N + i] - f[R_SY * N + i];
    f[R_DINIT * N + i] = sqrt(ddx * ddx + ddy * ddy);
}

__global__ __launch_bounds__(BLOCK) void k_set_state(const int n, double *f, const int32_t *idx, const int k,
                                                     const double *state) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    for (int r = 0; r < 6; ++r) f[r * (size_t)n + i] = state[6 * j + r];
}

__global__ __launch_bounds__(BLOCK) void k_set_max_steps(const int n, uint32_t *packed, const int32_t *idx, const int k,
                                                         const int32_t *maxs) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const int i = idx ? idx[j] : j;
    if (i < 0 || i >= n) return;
    const uint32_t p = packed[i];
    const int m = maxs[j] < 0 ? 0 : maxs[j];
    packed[i] = pk_make(pk_steps(p), (uint32_t)m, pk_stages(p));
}

__global__ __launch_bounds__(BLOCK) void k_get_episode(const int n, const uint32_t *packed, int32_t *steps,
                                                       int32_t *maxs) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = packed[i];
    if (steps) steps[i] = (int32_t)pk_steps(p);
    if (maxs) maxs[i] = (int32_t)pk_max(p);
}

template <bool PER_ENV>
__global__ __launch_bounds__(BLOCK) void k_observe(const KParams P, const int n, const double *f, const float *steering,
                                                   float *obs) {
    __shared__ __attribute__((aligned(16))) float tile[BLOCK * OBS];
    const int block_first = blockIdx.x * BLOCK;
    const int i = block_first + threadIdx.x;
    const bool valid = i < n;
    float of[OBS];
    if (valid) {
        const size_t N = (size_t)n;
        Goal g;
        if (PER_ENV) {
            g.gx = f[R_GX * N + i]; g.gy = f[R_GY * N + i]; g.sg = f[R_SG * N + i]; g.cg = f[R_CG * N + i];
        } else {
            g.gx = P.gx; g.gy = P.gy; g.sg = P.sg; g.cg = P.cg;
        }
        double s1, c1, s2, c2, sd = 0.0, cd = 1.0;
        sincos(f[R_PSI1 * N + i], &s1, &c1);
        sincos(f[R_PSI2 * N + i], &s2, &c2);
        if (steering) sincos((double)steering[i], &sd, &cd);
        observe(P, g, f[R_X1 * N + i], f[R_Y1 * N + i], f[R_X2 * N + i], f[R_Y2 * N + i], s1, c1, s2, c2, sd, cd, of);
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
    int n = 0, device = 0;
    tt_params params{};
    KParams kp{};
    double *f = nullptr;
    float *prev_steer = nullptr;
    uint32_t *packed = nullptr;
    bool per_env = false;
    uint64_t seed = 0, nonce = 0;
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

KParams make_kparams(const tt_params &p) {
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
    for (int i = 0; i < 3; ++i) { k.rlo[i] = p.reset_lo[i]; k.rhi[i] = p.reset_hi[i]; }
    k.extra_steps = p.extra_steps; k.fixed_max = p.fixed_max_steps;
    k.term_mask = p.term_mask;
    return k;
}

inline int grid_for(int n) { return (n + BLOCK - 1) / BLOCK; }

template <bool PER_ENV, bool INFO>
void launch_step(tt_env *e, bool auto_reset, const float *action, float *obs, float *reward, uint8_t *done,
                 const Info &info, hipStream_t s) {
    const dim3 g(grid_for(e->n)), b(BLOCK);
    if (auto_reset)
        hipLaunchKernelGGL((k_step<PER_ENV, INFO, true>), g, b, 0, s, e->kp, e->n, e->f, e->prev_steer, e->packed, action,
                           obs, reward, done, info, e->seed, e->nonce);
    else
        hipLaunchKernelGGL((k_step<PER_ENV, INFO, false>), g, b, 0, s, e->kp, e->n, e->f, e->prev_steer, e->packed, action,
                           obs, reward, done, info, e->seed, e->nonce);
}

}  // namespace

extern "C" {

int tt_version(void) { return TT_VERSION; }

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
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, TT_ENODEV, "tt_env_create: no HIP device visible");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= count) return fail(nullptr, TT_ENODEV, "tt_env_create: device %d of %d", device, count);
    tt_params p;
    if (params) p = *params;
    else tt_params_default(0, &p);
    if (!(p.L1 > 0.0) || !(p.L2 > 0.0) || !(p.dt > 0.0) || !(p.step_length > 0.0) || !(p.map_max_x > p.map_min_x) ||
        !(p.map_max_y > p.map_min_y))
        return fail(nullptr, TT_EINVAL, "tt_env_create: non-physical parameters");
    tt_env *e = new (std::nothrow) tt_env;
    if (!e) return fail(nullptr, TT_ENOMEM, "tt_env_create: host allocation failed");
    e->n = n_envs;
    e->device = device;
    e->params = p;
    e->kp = make_kparams(p);
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess) err = hipMalloc(&e->f, sizeof(double) * R_COUNT * (size_t)n_envs);
    if (err == hipSuccess) err = hipMalloc(&e->prev_steer, sizeof(float) * (size_t)n_envs);
    if (err == hipSuccess) err = hipMalloc(&e->packed, sizeof(uint32_t) * (size_t)n_envs);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(k_init, dim3(grid_for(n_envs)), dim3(BLOCK), 0, nullptr, e->kp, n_envs, e->f, e->prev_steer,
                           e->packed);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(nullptr);
    if (err != hipSuccess) {
        fail(nullptr, err == hipErrorOutOfMemory ? TT_ENOMEM : TT_EHIP, "tt_env_create: %s", hipGetErrorString(err));
        const int code = err == hipErrorOutOfMemory ? TT_ENOMEM : TT_EHIP;
        tt_env_destroy(e);
        return code;
    }
    *out = e;
    return TT_OK;
}

int tt_env_destroy(tt_env *env) {
    if (!env) return TT_OK;
    (void)hipSetDevice(env->device);
    if (env->f) (void)hipFree(env->f);
    if (env->prev_steer) (void)hipFree(env->prev_steer);
    if (env->packed) (void)hipFree(env->packed);
    delete env;
    return TT_OK;
}

int tt_env_num_envs(const tt_env *env) { return env ? env->n : TT_EINVAL; }

int tt_env_reset(tt_env *env, const uint8_t *mask, uint64_t seed, float *obs_out, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_reset: NULL handle");
    TT_HIP(env, hipSetDevice(env->device));
    env->seed = seed;
    env->nonce += 1;
    hipLaunchKernelGGL(k_reset, dim3(grid_for(env->n)), dim3(BLOCK), 0, stream, env->kp, env->n, env->f, env->packed, mask,
                       obs_out, env->seed, env->nonce);
    TT_HIP(env, hipGetLastError());
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
    hipLaunchKernelGGL(k_set_pose, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->kp, env->n, env->f, env->packed, idx, k,
                       start, goal, L2, obs_out);
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
    hipLaunchKernelGGL(k_set_attrs, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->n, env->f, idx, k, start, goal, L2);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_set_state(tt_env *env, const int32_t *idx, int k, const double *state, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_state: NULL handle");
    if (k < 0 || k > env->n || (k > 0 && !state))
        return fail(env, TT_EINVAL, "tt_env_set_state: k=%d outside [0,%d] or state NULL", k, env->n);
    if (k == 0) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    hipLaunchKernelGGL(k_set_state, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->n, env->f, idx, k, state);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_get_state(tt_env *env, double *state_out, tt_stream_t stream) {
    if (!env || !state_out) return fail(env, TT_EINVAL, "tt_env_get_state: NULL argument");
    TT_HIP(env, hipSetDevice(env->device));
    TT_HIP(env, hipMemcpyAsync(state_out, env->f, sizeof(double) * 6 * (size_t)env->n, hipMemcpyDeviceToDevice, stream));
    return TT_OK;
}

int tt_env_set_max_steps(tt_env *env, const int32_t *idx, int k, const int32_t *max_steps, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_set_max_steps: NULL handle");
    if (k < 0 || k > env->n || (k > 0 && !max_steps))
        return fail(env, TT_EINVAL, "tt_env_set_max_steps: k=%d outside [0,%d] or max_steps NULL", k, env->n);
    if (k == 0) return TT_OK;
    TT_HIP(env, hipSetDevice(env->device));
    hipLaunchKernelGGL(k_set_max_steps, dim3(grid_for(k)), dim3(BLOCK), 0, stream, env->n, env->packed, idx, k, max_steps);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_get_episode(tt_env *env, int32_t *steps, int32_t *max_steps, double *start, double *goal, double *L2,
                       tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_get_episode: NULL handle");
    TT_HIP(env, hipSetDevice(env->device));
    const size_t N = (size_t)env->n;
    if (steps || max_steps) {
        hipLaunchKernelGGL(k_get_episode, dim3(grid_for(env->n)), dim3(BLOCK), 0, stream, env->n, env->packed, steps,
                           max_steps);
        TT_HIP(env, hipGetLastError());
    }
    if (start)
        TT_HIP(env, hipMemcpyAsync(start, env->f + R_SX * N, sizeof(double) * 3 * N, hipMemcpyDeviceToDevice, stream));
    if (goal)
        TT_HIP(env, hipMemcpyAsync(goal, env->f + R_GX * N, sizeof(double) * 3 * N, hipMemcpyDeviceToDevice, stream));
    if (L2) TT_HIP(env, hipMemcpyAsync(L2, env->f + R_L2 * N, sizeof(double) * N, hipMemcpyDeviceToDevice, stream));
    return TT_OK;
}

int tt_env_observe(tt_env *env, const float *steering, float *obs_out, tt_stream_t stream) {
    if (!env || !obs_out) return fail(env, TT_EINVAL, "tt_env_observe: NULL argument");
    TT_HIP(env, hipSetDevice(env->device));
    const dim3 g(grid_for(env->n)), b(BLOCK);
    if (env->per_env)
        hipLaunchKernelGGL(k_observe<true>, g, b, 0, stream, env->kp, env->n, env->f, steering, obs_out);
    else
        hipLaunchKernelGGL(k_observe<false>, g, b, 0, stream, env->kp, env->n, env->f, steering, obs_out);
    TT_HIP(env, hipGetLastError());
    return TT_OK;
}

int tt_env_step(tt_env *env, const float *action, float *obs, float *reward, uint8_t *done, const tt_info *info,
                int auto_reset, tt_stream_t stream) {
    if (!env) return fail(nullptr, TT_EINVAL, "tt_env_step: NULL handle");
    if (!action || !obs || !reward || !done)
        return fail(env, TT_EINVAL, "tt_env_step: action, obs, reward and done are required");
    TT_HIP(env, hipSetDevice(env->device));
    Info ki{nullptr, nullptr, nullptr};
    bool want_info = false;
    if (info) {
        ki.comp = info->comp; ki.violation = info->violation; ki.flags = info->flags;
        want_info = ki.comp || ki.violation || ki.flags;
    }
    env->nonce += 1;
    if (env->per_env) {
        if (want_info) launch_step<true, true>(env, auto_reset != 0, action, obs, reward, done, ki, stream);
        else launch_step<true, false>(env, auto_reset != 0, action, obs, reward, done, ki, stream);
    } else {
        if (want_info) launch_step<false, true>(env, auto_reset != 0, action, obs, reward, done, ki, stream);
        else launch_step<false, false>(env, auto_reset != 0, action, obs, reward, done, ki, stream);
    }
    TT_HIP(env, hipGetLastError());
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
