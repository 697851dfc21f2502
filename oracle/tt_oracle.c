/* ORACLE (test infrastructure, not product) -- plain-C scalar restatement of the
 * truck-trailer backing environment step of pain7576/ddpg-trucktrailer.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may link or
 * call this; the product (ddpg-trucktrailer_amd/) never does.
 *
 * It follows the reference's formulas literally (libm sin/cos/atan2 on every angle, no
 * algebraic shortcuts), so that it is an independent check of the HIP kernel, which does
 * take shortcuts.  The one deliberate difference from the reference: the ODE is advanced
 * by ONE fixed Dormand-Prince step of h = dt in f64 instead of scipy's adaptive RK45
 * driver (the integrator the HIP kernel runs; BASELINE.md §2: within 2.3e-10 of scipy).
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against fixtures F1-F4
 * (tests/golden/, generated from the real reference), tolerance 1e-5 free-running.
 *
 * Reference sections restated (paths relative to the reference repo):
 *   tto_params_default   truck_trailer_sim/simv2.py:23-101   (simv1.py:23-99 for variant 1)
 *   tto_place            simv2.py:459-498, 263-267 and the callers' pose override DDPG/test.py:96-115
 *   rhs                  simv2.py:269-303
 *   integrate            scipy/integrate/_ivp/rk.py RK45 tableau (A, B, C), one step
 *   tto_observe          simv2.py:103-181
 *   reward               truck_trailer_sim/reward_functionv1.py:6-109 (carry), 144-506 (components)
 *   flags / done         simv2.py:305-345, 528-541; reward_functionv1.py:120-124
 */
#include "tt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846
#define DEG (PI / 180.0)

int tto_env_size(void) { return (int)sizeof(tto_env); }

void tto_params_default(tto_params *p, int variant) {
    memset(p, 0, sizeof(*p));
    p->L1 = variant ? 5.74 : 5.0;
    p->L2 = variant ? 10.192 : 7.0;
    p->hitch_offset = 0.0;
    p->v1x = -5.012;
    p->dt = 0.08;
    p->map_min = -40.0;
    p->map_max = 40.0;
    p->max_steer = 45.0 * DEG;
    p->max_expected_distance = sqrt(80.0 * 80.0 + 80.0 * 80.0);
    p->position_threshold = 0.5;
    p->orientation_threshold = 15.0 * DEG;
    p->step_length = 0.40096;
    p->extra_steps = 75;
    p->fixed_max_steps = variant ? 300 : 0;
    p->term_mask = variant ? 0x0F : 0x3F;
    p->variant = variant;
    p->stateless_reward = variant ? 1 : 0;
    p->goal[0] = 0.0;
    p->goal[1] = -30.0;
    p->goal[2] = 90.0 * DEG;
}

static double dist2goal(const tto_env *e) {
    double dx = e->y[4] - e->goal[0], dy = e->y[5] - e->goal[1];
    return sqrt(dx * dx + dy * dy);
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

void tto_observe(const tto_params *p, const tto_env *e, double steering, float *obs) {
    const double psi1 = e->y[0], psi2 = e->y[1], x1 = e->y[2], y1 = e->y[3], x2 = e->y[4], y2 = e->y[5];
    const double gx = e->goal[0], gy = e->goal[1], gyaw = e->goal[2];
    const double centre = (p->map_max + p->map_min) / 2.0, half = (p->map_max - p->map_min) / 2.0;
    const double M = p->max_expected_distance;
    double dxg = gx - x2, dyg = gy - y2;
    double dist = sqrt((x2 - gx) * (x2 - gx) + (y2 - gy) * (y2 - gy));
    double to_goal = atan2(dyg, dxg);
    double d_long = dxg * cos(psi2) + dyg * sin(psi2);
    double d_lat = -dxg * sin(psi2) + dyg * cos(psi2);
    double hitch = psi1 - psi2, ori_err = gyaw - psi2, head_err = to_goal - (psi2 + 180.0 * DEG);
    double o[TTO_OBS_DIM] = {
        (x1 - centre) / half, (y1 - centre) / half, sin(psi1), cos(psi1),
        (x2 - centre) / half, (y2 - centre) / half, sin(psi2), cos(psi2),
        sin(hitch), cos(hitch), sin(steering), cos(steering),
        (gx - centre) / half, (gy - centre) / half, sin(gyaw), cos(gyaw),
        clampd(dist / M, 0.0, 1.0), clampd(d_long / M, -1.0, 1.0), clampd(d_lat / M, -1.0, 1.0),
        sin(ori_err), cos(ori_err), sin(head_err), cos(head_err)};
    for (int i = 0; i < TTO_OBS_DIM; i++) obs[i] = (float)o[i];
}

static int compute_max_steps(const tto_params *p, const tto_env *e) {
    if (p->fixed_max_steps > 0) return p->fixed_max_steps;
    double dx = e->goal[0] - e->start[0], dy = e->goal[1] - e->start[1];
    return (int)(sqrt(dx * dx + dy * dy) / p->step_length) + p->extra_steps;
}

void tto_place(const tto_params *p, tto_env *e, const double start[3], const double goal[3], double L2, float *obs) {
    memset(e, 0, sizeof(*e));
    memcpy(e->start, start, sizeof(e->start));
    memcpy(e->goal, goal ? goal : p->goal, sizeof(e->goal));
    e->L2 = L2 > 0.0 ? L2 : p->L2;
    double yaw = start[2];
    /* the fresh state is stored as float32 (simv2.py:489) */
    e->y[0] = (double)(float)yaw;
    e->y[1] = (double)(float)yaw;
    e->y[2] = (double)(float)(start[0] + e->L2 * cos(yaw));
    e->y[3] = (double)(float)(start[1] + e->L2 * sin(yaw));
    e->y[4] = (double)(float)start[0];
    e->y[5] = (double)(float)start[1];
    e->max_steps = compute_max_steps(p, e);
    if (obs) tto_observe(p, e, 0.0, obs);
}

void tto_set_state(tto_env *e, const double y[6]) { memcpy(e->y, y, sizeof(e->y)); }

static void rhs(const tto_params *p, double L2, double steering, const double *y, double *d) {
    double hitch = y[0] - y[1];
    double w1 = (p->v1x / p->L1) * tan(steering);
    double v2 = p->v1x * cos(hitch) + p->hitch_offset * w1 * sin(hitch);
    double w2 = (p->v1x / L2) * sin(hitch) - (p->hitch_offset / L2) * w1 * cos(hitch);
    d[0] = w1;
    d[1] = w2;
    d[2] = p->v1x * cos(y[0]);
    d[3] = p->v1x * sin(y[0]);
    d[4] = v2 * cos(y[1]);
    d[5] = v2 * sin(y[1]);
}

/* the ODE right-hand side alone (simv2.py:269-303 == simv1.py:180-214): checked against the reference's kinematic_model
 * by fixture F6 */
void tto_rhs(const tto_params *p, double L2, double steering, const double y[6], double d[6]) { rhs(p, L2, steering, y, d); }

static void integrate(const tto_params *p, double L2, double steering, double *y) {
    static const double A[6][5] = {{0},
                                   {1.0 / 5},
                                   {3.0 / 40, 9.0 / 40},
                                   {44.0 / 45, -56.0 / 15, 32.0 / 9},
                                   {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
                                   {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
    static const double B[6] = {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
    double k[6][6], yt[6];
    const double h = p->dt;
    for (int s = 0; s < 6; s++) {
        for (int c = 0; c < 6; c++) {
            double acc = 0.0;
            for (int j = 0; j < s; j++) acc += A[s][j] * k[j][c];
            yt[c] = y[c] + h * acc;
        }
        rhs(p, L2, steering, yt, k[s]);
    }
    for (int c = 0; c < 6; c++) {
        double acc = 0.0;
        for (int j = 0; j < 6; j++) acc += B[j] * k[j][c];
        y[c] += h * acc;
    }
}

static double reward(const tto_params *p, tto_env *e, const float *obs, double *info) {
    const double gx = e->goal[0], gy = e->goal[1];
    const double cur = dist2goal(e);
    const double init = sqrt((gx - e->start[0]) * (gx - e->start[0]) + (gy - e->start[1]) * (gy - e->start[1])) + 1e-6;
    const int rmax = (int)(init / p->step_length) + p->extra_steps;
    const float steer_now = atan2f(obs[10], obs[11]);

    if (!e->has_carry || p->stateless_reward) { /* first step of an episode: reward_functionv1.py:40-76;
                                                   simv1.py:435 builds the reward afresh every step */
        e->prev_dist = cur;
        e->prev_steer = steer_now;
        e->cum_back = 0.0;
        e->bt_steps = 0;
        for (int i = 0; i < 5; i++) e->hist[i] = cur;
        e->stages[0] = e->stages[1] = e->stages[2] = 0;
        e->closest = cur;
        e->has_carry = 1;
    } else if (cur < e->closest) {
        e->closest = cur;
    }

    /* :189-238 */
    double jp = clampd((init - cur) / init, 0.0, 1.0);
    double w_orient = (tanh(7.0 * (jp - 0.3)) + 1.0) / 2.0;
    double w_head = 1.0 - w_orient;

    /* :144-187 */
    double inst = e->prev_dist - cur;
    double prog = inst > 0 ? tanh(inst) : tanh(inst) * 0.5;
    memmove(e->hist, e->hist + 1, 4 * sizeof(double));
    e->hist[4] = cur;
    double prog_net = tanh((e->hist[0] - cur) / 2.0) * 0.5;
    int mono = e->hist[2] >= e->hist[3] && e->hist[3] >= e->hist[4];
    double progress = prog + prog_net + (mono ? 0.2 : 0.0);

    /* :285-309 */
    double want = atan2(gy - e->y[5], gx - e->y[4]);
    float have = atan2f(obs[6], obs[7]);
    double diff = want - ((double)have + 180.0 * DEG);
    diff = atan2(sin(diff), cos(diff));
    double heading = cos(diff);

    /* :311-324, float32 product first (numpy keeps f32 * python-float in f32) */
    float orient15 = obs[20] * 15.0f;

    /* :338-367 */
    float ori_err = fabsf(atan2f(obs[19], obs[20]));
    double staged = 0.0;
    if (cur <= 5.0) {
        staged += 10.0;
        e->stages[0] = 1;
    }
    if (cur <= 2.0 && (double)ori_err <= 45.0 * DEG && !e->stages[1]) {
        staged += 25.0;
        e->stages[1] = 1;
    }
    if (cur <= p->position_threshold && (double)ori_err <= p->orientation_threshold && !e->stages[2]) {
        staged += 100.0;
        e->stages[2] = 1;
    }

    /* :369-421 */
    double safety = 0.0;
    int viol = 0;
    double hitch = fabs(e->y[0] - e->y[1]);
    if (hitch > 85.0 * DEG) {
        safety += -500.0;
        viol = 1;
    } else if (hitch > 70.0 * DEG) {
        safety += -50.0;
        viol = 2;
    }
    int major = 0, minor = 0;
    for (int i = 2; i < 6; i++) {
        double v = e->y[i];
        if (v < p->map_min - 2 || v > p->map_max + 2) major = 1;
        if (v < p->map_min || v > p->map_max) minor = 1;
    }
    if (major) {
        safety += -500.0;
        viol = 3;
    } else if (minor) {
        safety += -50.0;
        viol = 4;
    }
    if (gy > e->y[5]) {
        safety += -500.0;
        viol = 5;
    }
    if (e->steps >= rmax) {
        safety += -500.0;
        viol = 6;
    }
    int excessive = cur > e->closest + 6.0;
    if (excessive) {
        safety += -500.0;
        viol = 7;
    }

    /* :423-439 */
    double explore = e->steps < rmax * 0.5 ? 4.0 : (e->steps < rmax * 0.8 ? 2.0 : 0.0);

    /* :240-283 */
    double back_step = cur - e->prev_dist;
    e->cum_back += back_step > 0 ? back_step : 0.0;
    e->bt_steps += 1;
    double tf = e->bt_steps / 50.0;
    double budget = 5.0 * (tf < 1.0 ? tf : 1.0);
    double excess = e->cum_back - budget;
    if (excess < 0) excess = 0;
    double back = excess > 0 ? -(pow(excess, 1.5) * 0.5) : 0.0;

    /* :326-335, previous_steering is never refreshed */
    double smooth = (double)fabsf(steer_now - e->prev_steer) / (90.0 * DEG);

    int success = cur <= p->position_threshold && (double)ori_err <= p->orientation_threshold;
    double final_bonus = success ? 200.0 : 0.0;
    e->prev_dist = cur;

    double c_prog = progress * 15.0 * 1.0;
    double c_head = heading * 15.0 * w_head;
    double c_orient = (double)orient15 * w_orient;
    double c_back = back * 1.0;
    double c_smooth = smooth * -25.0;
    double total = 0.0 + c_prog + c_head + c_orient + staged + safety + explore + c_back + c_smooth + final_bonus;

    e->violation = (uint8_t)viol;
    e->flags = (uint8_t)((excessive ? TTO_F_EXCESSIVE_BACK : 0) | (success ? TTO_F_SUCCESS : 0));
    if (info) {
        info[TTO_I_TOTAL] = total;
        info[TTO_I_PROGRESS] = c_prog;
        info[TTO_I_HEADING] = c_head;
        info[TTO_I_ORIENT] = c_orient;
        info[TTO_I_STAGED] = staged;
        info[TTO_I_SAFETY] = safety;
        info[TTO_I_EXPLORE] = explore;
        info[TTO_I_FINAL] = final_bonus;
        info[TTO_I_BACKWARD] = c_back;
        info[TTO_I_SMOOTH] = c_smooth;
        info[TTO_I_CUMBACK] = e->cum_back;
        info[TTO_I_BUDGET] = budget;
    }
    return total;
}

void tto_step(const tto_params *p, tto_env *e, float action, float *obs, double *rew, uint8_t *done, double *info) {
    /* simv2.py:504-505: clip in f64 against np.radians(45) */
    double steering = clampd((double)action, -p->max_steer, p->max_steer);
    integrate(p, e->L2, steering, e->y);
    float o[TTO_OBS_DIM];
    tto_observe(p, e, steering, o);
    e->steps += 1;
    double total = reward(p, e, o, info);

    uint8_t f = e->flags;
    if (fabs(e->y[0] - e->y[1]) > 90.0 * DEG) f |= TTO_F_JACKKNIFE;
    for (int i = 2; i < 6; i++)
        if (e->y[i] < p->map_min || e->y[i] > p->map_max) f |= TTO_F_OUT_OF_MAP;
    if (e->steps >= e->max_steps) f |= TTO_F_MAX_STEPS;
    if (dist2goal(e) <= p->position_threshold && (double)fabsf(atan2f(o[19], o[20])) <= p->orientation_threshold)
        f |= TTO_F_GOAL_REACHED;
    if (e->goal[1] > e->y[5]) f |= TTO_F_GOAL_PASSED;
    e->flags = f;
    if (obs) memcpy(obs, o, sizeof(o));
    if (rew) *rew = total;
    if (done) *done = (f & p->term_mask) ? 1 : 0;
}

void tto_step_batch(const tto_params *p, tto_env *envs, int n, const float *actions, float *obs, double *rew,
                    uint8_t *done, double *info, int nthreads) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
    for (int i = 0; i < n; i++)
        tto_step(p, &envs[i], actions[i], obs ? obs + (size_t)i * TTO_OBS_DIM : 0, rew ? rew + i : 0,
                 done ? done + i : 0, info ? info + (size_t)i * TTO_NINFO : 0);
    (void)nthreads;
}

void tto_place_batch(const tto_params *p, tto_env *envs, int n, const double *start, const double *goal,
                     const double *L2, float *obs) {
    for (int i = 0; i < n; i++)
        tto_place(p, &envs[i], start + 3 * (size_t)i, goal ? goal + 3 * (size_t)i : 0, L2 ? L2[i] : 0.0,
                  obs ? obs + (size_t)i * TTO_OBS_DIM : 0);
}

static inline uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double uni(uint64_t *s, double lo, double hi) {
    return lo + (hi - lo) * ((double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0));
}
static void random_place(const tto_params *p, tto_env *e, uint64_t *s) {
    /* simv2.py:331-337 draw order x, y, yaw */
    double st[3];
    st[0] = uni(s, -27.0, 27.0);
    st[1] = uni(s, 0.0, 27.0);
    st[2] = uni(s, 45.0 * DEG, 120.0 * DEG);
    tto_place(p, e, st, 0, 0.0, 0);
}

long tto_rollout_random(const tto_params *p, int n_envs, int n_steps, uint64_t seed, int nthreads, double *reward_sum) {
    double total = 0.0;
    long count = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static) reduction(+ : total, count)
#endif
    for (int i = 0; i < n_envs; i++) {
        uint64_t s = seed * 0x100000001B3ull + (uint64_t)i;
        tto_env e;
        float obs[TTO_OBS_DIM];
        random_place(p, &e, &s);
        for (int t = 0; t < n_steps; t++) {
            float a = (float)(uni(&s, -1.0, 1.0) * (PI / 4));
            double r;
            uint8_t d;
            tto_step(p, &e, a, obs, &r, &d, 0);
            total += r;
            count++;
            if (d) random_place(p, &e, &s);
        }
    }
    (void)nthreads;
    if (reward_sum) *reward_sum = total;
    return count;
}
