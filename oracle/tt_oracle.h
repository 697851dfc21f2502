/* ORACLE (test infrastructure, not product).  Plain-C scalar restatement of the
 * reference's simv2 step: see tt_oracle.c for the per-function reference citations.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it. */
#ifndef TT_ORACLE_H
#define TT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTO_OBS_DIM 23
/* info vector layout (reward_functionv1.py:489-504) */
enum { TTO_I_TOTAL = 0, TTO_I_PROGRESS, TTO_I_HEADING, TTO_I_ORIENT, TTO_I_STAGED, TTO_I_SAFETY,
       TTO_I_EXPLORE, TTO_I_FINAL, TTO_I_BACKWARD, TTO_I_SMOOTH, TTO_I_CUMBACK, TTO_I_BUDGET, TTO_NINFO };
/* flag bits (simv2.py:528-541) */
enum { TTO_F_JACKKNIFE = 1, TTO_F_OUT_OF_MAP = 2, TTO_F_MAX_STEPS = 4, TTO_F_GOAL_REACHED = 8,
       TTO_F_GOAL_PASSED = 16, TTO_F_EXCESSIVE_BACK = 32, TTO_F_SUCCESS = 64 };

typedef struct {
    double L1, L2, hitch_offset, v1x, dt;
    double map_min, map_max;
    double max_steer;
    double max_expected_distance;
    double position_threshold, orientation_threshold;
    double step_length;
    int32_t extra_steps;
    int32_t fixed_max_steps; /* 0: derive from the start distance (simv2); 300 for simv1 */
    uint32_t term_mask;
    int32_t variant;
    int32_t stateless_reward; /* simv1.py:435: nothing carried between steps */
    int32_t reserved_;
    double goal[3];
} tto_params;

typedef struct {
    double y[6];     /* psi1 psi2 x1 y1 x2 y2 */
    double start[3]; /* startx starty startyaw */
    double goal[3];
    double L2;
    double prev_dist, cum_back, closest, hist[5];
    float prev_steer;
    int32_t steps, max_steps, bt_steps;
    uint8_t stages[3];
    uint8_t has_carry;
    uint8_t flags, violation;
} tto_env;

int tto_env_size(void);
void tto_params_default(tto_params *p, int variant);
void tto_place(const tto_params *p, tto_env *e, const double start[3], const double goal[3], double L2, float *obs);
void tto_set_state(tto_env *e, const double y[6]);
void tto_rhs(const tto_params *p, double L2, double steering, const double y[6], double d[6]);
void tto_observe(const tto_params *p, const tto_env *e, double steering, float *obs);
void tto_step(const tto_params *p, tto_env *e, float action, float *obs, double *reward, uint8_t *done, double *info);
void tto_step_batch(const tto_params *p, tto_env *envs, int n, const float *actions, float *obs, double *reward,
                    uint8_t *done, double *info, int nthreads);
void tto_place_batch(const tto_params *p, tto_env *envs, int n, const double *start, const double *goal,
                     const double *L2, float *obs);
/* bounded CPU-baseline workload: uniform-random steering, reset on done; returns env-steps executed */
long tto_rollout_random(const tto_params *p, int n_envs, int n_steps, uint64_t seed, int nthreads, double *reward_sum);

#ifdef __cplusplus
}
#endif
#endif
