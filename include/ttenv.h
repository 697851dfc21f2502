/* ttenv.h -- C ABI of libttenv.so: the batched truck-trailer backing environment on MI355X.
 *
 * The reference (pain7576/ddpg-trucktrailer) has no FFI; the interface its callers use for
 * this path is the gym-style Python surface of `Truck_trailer_Env_2`
 * (truck_trailer_sim/simv2.py:20-101, 459-545) plus the public attributes they read and
 * write (DDPG/trainv2.py:488-531, DDPG/test.py:96-115, DDPG/heatmap.py:79-168).  Each entry
 * point below names the reference interface it replaces; INTEGRATION.md shows the ctypes
 * stub that binds them behind that Python surface.
 *
 * Conventions
 *  - plain C: opaque handle, plain pointers and sizes, no C++/torch types;
 *  - every call returns 0 (TT_OK) or a negative TT_E* code and never throws;
 *    tt_last_error() returns a message owned by the handle (or by the library for a NULL handle);
 *  - every array pointer is a DEVICE pointer owned by the caller (e.g. torch `data_ptr()`),
 *    layouts as stated per argument; N = number of envs of the handle;
 *  - work is enqueued on the caller's stream (`tt_stream_t` is `hipStream_t`; NULL = the
 *    default stream) and is asynchronous; nothing here synchronises or allocates after create;
 *  - one host thread per handle; handles are independent (one process per GPU, one handle per
 *    process is the multi-GPU model).
 */
#ifndef TTENV_H
#define TTENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_VERSION 3
#define TT_OBS_DIM 23 /* simv2.py:76 */
#define TT_MAX_EPISODE_STEPS 4095 /* steps and max_episode_steps are 12-bit packed counters; larger values are TT_EINVAL */

enum { TT_OK = 0, TT_EINVAL = -1, TT_ENOMEM = -2, TT_EHIP = -3, TT_ENODEV = -4 };

/* termination causes, one bit each, in the order of simv2.py:528-541; bit 6 = reward-side success */
enum {
    TT_F_JACKKNIFE = 1, TT_F_OUT_OF_MAP = 2, TT_F_MAX_STEPS = 4, TT_F_GOAL_REACHED = 8,
    TT_F_GOAL_PASSED = 16, TT_F_EXCESSIVE_BACK = 32, TT_F_SUCCESS = 64
};
#define TT_TERM_SIMV2 0x3F /* all six causes end an episode (simv2.py:541) */
#define TT_TERM_SIMV1 0x0F /* jackknife | out of map | max steps | goal (simv1.py:432) */

/* `violation_type` of reward_functionv1.py:378-419, last writer wins */
enum {
    TT_V_NONE = 0, TT_V_JACKKNIFE, TT_V_JACKKNIFE_WARNING, TT_V_MAJOR_BOUNDARY, TT_V_MINOR_BOUNDARY,
    TT_V_PAST_THE_GOAL, TT_V_MAX_STEP, TT_V_EXCESSIVE_BACKWARD
};

/* rows of tt_info.comp, the reward_info dict of reward_functionv1.py:489-504 */
enum {
    TT_I_TOTAL = 0, TT_I_PROGRESS, TT_I_HEADING, TT_I_ORIENT, TT_I_STAGED, TT_I_SAFETY, TT_I_EXPLORE,
    TT_I_FINAL, TT_I_BACKWARD, TT_I_SMOOTH, TT_I_CUMBACK, TT_I_BUDGET, TT_NINFO
};

typedef struct ihipStream_t *tt_stream_t; /* == hipStream_t */
typedef struct tt_env tt_env;

/* Constants of Truck_trailer_Env_2.__init__ (simv2.py:23-101); variant 1 = simv1.py:23-99. */
typedef struct tt_params {
    double L1, L2;             /* wheelbase, trailer length (L2 is the default; per-env via tt_env_set_pose) */
    double hitch_offset, v1x;  /* 0.0, -5.012 */
    double dt;                 /* 0.08: one fixed Dormand-Prince step per env step */
    double map_min_x, map_max_x, map_min_y, map_max_y; /* -40..40 */
    double max_steer;          /* np.radians(45) */
    double position_threshold, orientation_threshold; /* 0.5, deg2rad(15) */
    double step_length;        /* 0.40096 (simv2.py:265) */
    int32_t extra_steps;       /* 75 */
    int32_t fixed_max_steps;   /* 0 = int(d0/step_length)+extra_steps (simv2); 300 (simv1.py:95) */
    uint32_t term_mask;        /* which TT_F_* bits end an episode */
    int32_t variant;           /* 0 simv2, 1 simv1 */
    int32_t stateless_reward;  /* 0: the reward carry persists over the episode (simv2.py:347-373);
                                  1: a fresh RewardFunction every step, nothing carried (simv1.py:435) */
    int32_t reserved_;
    double goal[3];            /* default goal x, y, yaw: 0, -30, pi/2 (simv2.py:335-337) */
    double reset_lo[3], reset_hi[3]; /* start x, y, yaw ~ U(lo, hi) (simv2.py:331-333) */
} tt_params;

/* Optional per-step detail: the `info` dict of env.step (reward_functionv1.py:489-504) and the
 * public flags callers read afterwards (heatmap.py:159-168).  Any member may be NULL. */
typedef struct tt_info {
    double *comp;       /* [TT_NINFO, N] f64, row-major by component (row TT_I_TOTAL = f64 reward) */
    uint8_t *violation; /* [N] TT_V_* */
    uint8_t *flags;     /* [N] TT_F_* bits, before masking with term_mask */
} tt_info;

int tt_version(void);
const char *tt_last_error(const tt_env *env);

/* Truck_trailer_Env_2() / Truck_trailer_Env_1(): fills `out` with the reference constants. */
int tt_params_default(int variant, tt_params *out);

/* Truck_trailer_Env_2.__init__ for N envs on HIP device `device` (simv2.py:23-101). */
int tt_env_create(int n_envs, int device, const tt_params *params, tt_env **out);
int tt_env_destroy(tt_env *env);
int tt_env_num_envs(const tt_env *env);

/* env.reset(seed) (simv2.py:459-498) for the envs whose mask byte is non-zero (NULL = all):
 * start pose ~ reset_lo/hi from counter-based Philox keyed by (seed, env, episode number) (distributional
 * parity with np.random.seed; the bit-exact seeded pose is tt_env_set_pose's job).  A full reset (mask NULL)
 * restarts every env at episode 0, so the poses are a pure function of `seed`; state stored
 * f32-rounded, step counter and reward carry cleared, obs rows written with steering 0.
 * Also fixes the seed used by tt_env_step's auto-reset.  obs_out [N,23] f32 may be NULL. */
int tt_env_reset(tt_env *env, const uint8_t *mask, uint64_t seed, float *obs_out, tt_stream_t stream);

/* Start-pose pool for resets (simv1.py:255-282 draws poses by rejection sampling against a Dubins-path
 * feasibility test, which is host logic): when a pool of m >= 1 poses [m,3] f64 (x, y, yaw) is set, tt_env_reset
 * and the in-kernel auto-reset draw uniformly from it instead of from reset_lo/hi.  The pool stays owned by the
 * caller and must outlive its use; m = 0 (or pool NULL) goes back to the box distribution. */
int tt_env_set_reset_pool(tt_env *env, const double *pool, int m);

/* The callers' pose-override pattern (DDPG/test.py:96-115, heatmap.py:79-122): for j < k set
 * env idx[j] (idx NULL = env j) to start[j] = (startx, starty, startyaw), optional goal[j] and
 * L2[j], state = trailer pose with the truck L2 ahead rounded to f32, max_episode_steps =
 * compute_max_steps(), episode cleared; writes obs rows idx[j] of obs_out [N,23] if non-NULL. */
int tt_env_set_pose(tt_env *env, const int32_t *idx, int k, const double *start /*[k,3]*/,
                    const double *goal /*[k,3] or NULL*/, const double *L2 /*[k] or NULL*/, float *obs_out,
                    tt_stream_t stream);

/* Plain attribute writes `env.startx/starty/startyaw = ...`, `env.goalx/goaly/goalyaw = ...`,
 * `env.L2 = ...` (heatmap.py:79-89): like tt_env_set_pose's arguments (each may be NULL = keep)
 * but WITHOUT touching the kinematic state, the step counter or the reward carry. */
int tt_env_set_attrs(tt_env *env, const int32_t *idx, int k, const double *start /*[k,3] or NULL*/,
                     const double *goal /*[k,3] or NULL*/, const double *L2 /*[k] or NULL*/, tt_stream_t stream);

/* `env.state = ...` / `env.state` (simv2.py:489, trainv2.py:499,522): raw f64 kinematic state
 * psi1, psi2, x1, y1, x2, y2.  set: state [k,6] row-major for envs idx[j]; get: [6,N] SoA. */
int tt_env_set_state(tt_env *env, const int32_t *idx, int k, const double *state, tt_stream_t stream);
int tt_env_get_state(tt_env *env, double *state_out, tt_stream_t stream);

/* `env.max_episode_steps = ...` (heatmap.py:96) for envs idx[j].  Values outside [0, TT_MAX_EPISODE_STEPS] are refused
 * with TT_EINVAL and nothing is written; to check them this call copies max_steps to the host and waits for `stream`
 * (a setter, not a hot call; not capturable). */
int tt_env_set_max_steps(tt_env *env, const int32_t *idx, int k, const int32_t *max_steps, tt_stream_t stream);

/* `env.episode_steps = ...` (simv2.py:94, 523-530; written by a caller at DDPG/episode_replay_collectorv2.py:269) for envs
 * idx[j]: the step counter that the exploration tiers, the max-step penalty and `max_steps_reached` read.  Like the
 * reference's attribute write it leaves the reward carry alone (reward_state: previous distance, backward-movement count,
 * stage latches -- only reset / tt_env_set_pose clear it).  Range-checked and synchronising like tt_env_set_max_steps. */
int tt_env_set_steps(tt_env *env, const int32_t *idx, int k, const int32_t *steps, tt_stream_t stream);

/* Episode bookkeeping read-back; any pointer may be NULL.  steps/max_steps [N] i32;
 * start, goal [3,N] f64 (startx.., goalx..: trainv2.py:503-508); L2 [N] f64. */
int tt_env_get_episode(tt_env *env, int32_t *steps, int32_t *max_steps, double *start, double *goal, double *L2,
                       tt_stream_t stream);

/* env.compute_observation(env.state, steering) (simv2.py:103-181) for all envs;
 * steering [N] f32 radians or NULL for 0.  obs_out [N,23] f32 row-major. */
int tt_env_observe(tt_env *env, const float *steering, float *obs_out, tt_stream_t stream);

/* env.step(action) for all N envs (simv2.py:499-545): clip to +-max_steer, one DP5 step of the
 * kinematic ODE in f64, 23-dim observation, reward_functionv1 reward with its carry, flags.
 *   action [N] f32 radians; obs [N,23] f32 row-major; reward [N] f32; done [N] u8;
 *   info optional (NULL or members NULL).
 * auto_reset != 0: an env that is done is re-placed like tt_env_reset (its obs row is then the
 * fresh episode's first observation; reward/done/info still describe the finished step).  The
 * reference never resets by itself (trainv2.py:489); auto_reset = 0 reproduces that. */
/* Optional device-side count of vector steps: every tt_env_step / tt_env_step_random launch adds 1 to *counter
 * (device int64, caller-owned; NULL detaches).  The trajectory ring's sampler (tt_ring_sample: k_dev) reads it, so a
 * vector step needs no separate launch to advance the ring. */
int tt_env_set_step_counter(tt_env *env, int64_t *counter);

int tt_env_step(tt_env *env, const float *action, float *obs, float *reward, uint8_t *done, const tt_info *info,
                int auto_reset, tt_stream_t stream);

/* Ring addressing: the trajectory ring of a rollout loop (obs [slots,N,23], act / rew [slots,N] f32, done [slots,N] u8) and a
 * device cursor {t, t+1, t-1, t > 0} (slot numbers of the running vector step, written by the step's opening launch:
 * tt_ring_cursor of tt_mlp_split_pack[_and_sample]).  Launches that take a view read / write the step's slots through the
 * cursor instead of through per-slot pointers, so ONE captured hipGraph serves every ring position. */
typedef struct tt_ring_view {
    int32_t *cursor;        /* [TT_CURSOR_INTS] device ([12..16]: the hand-over words, below): [4..7] / [8..11] the cursors of even / odd steps (written by the opening launch of
                               the step), [0..3] the running step's copy, left by tt_actor_act_ring for tt_env_step_ring */
    float *obs, *act, *rew;
    uint8_t *done;
    int32_t n_envs, slots;
} tt_ring_view;
typedef struct tt_ring_cursor {
    const int64_t *k_dev;   /* vector steps completed (tt_env_set_step_counter); nothing may advance it beside the launch */
    int32_t slots, reserved_;
    int32_t *cursor;        /* [TT_CURSOR_INTS] device (tt_ring_view): the launch writes the four numbers of step *k_dev at [4 + 4 (k & 1)] */
} tt_ring_cursor;
/* Image hand-over (cursor[12..15], zero-initialised by the caller, re-zeroed when *k_dev is set back): a pack launch given a
 * cursor ends by publishing "cursor and image of step k = *k_dev are complete" as cursor[12 + (k & 1)] = k + 1 (release,
 * device scope), and tt_actor_act_ring begins by waiting for cursor[12 + (k & 1)] >= k + 1 with k = its *step_dev (bounded:
 * after 0.25 s it sets cursor[15] = k + 1 and goes on -- TT_CURSOR_GAVE_UP; a caller that lets the two launches run
 * unordered checks that word).  The two launches of a step therefore need NO stream / graph dependency between them; the
 * pack launch of step k + 2, which overwrites the same image, must still be ordered behind the policy launch of step k. */
#define TT_CURSOR_INTS 32
#define TT_CURSOR_GAVE_UP 15
/* cursor[18..19] (optional, 0 = none): the 64-bit address of ONE int of device-visible host memory (hipHostMalloc; a torch
 * pinned tensor) that a launch which gives up sets as well (system scope), so that the host sees a give-up by reading its own
 * memory, without a copy or a synchronize.  The caller writes the address whenever it zeroes the cursor buffer. */
#define TT_CURSOR_GAVE_UP_MIRROR 18
/* Step-chain progress (cursor[16]): tt_actor_act_ring of step k begins by storing k + 1 there (device scope).  A launch starts
 * only when everything in front of it on its stream is complete and written back, so cursor[16] >= k + 1 says: the env step of
 * step k - 1 -- and every launch before it -- is over and visible.  tt_mlp_forward_multi_sampled can wait for that word instead of
 * for a stream / graph dependency on the env step (tt_sample_args.step_progress), so that a loop's learn chain needs no edge
 * from its step chain either.  The caller sets the word to *k_dev whenever it sets *k_dev (resume). */
#define TT_CURSOR_PROGRESS 16
/* tt_env_step with obs -> ring slot t+1, reward and done -> slot t (env.step of the vector loop, trainv2.py:520-525). */
int tt_env_step_ring(tt_env *env, const float *action, const tt_ring_view *ring, int auto_reset, tt_stream_t stream);

/* tt_env_step with BASELINE.json config 2's "random policy" drawn inside the kernel:
 * action = U(-1,1) * pi/4 from Philox keyed by (policy_seed, env, step-in-episode, episode number), so no
 * host-side counter changes between launches and a captured hipGraph of K steps replays correctly.
 * action_out [N] f32 (the drawn steering, e.g. for a replay buffer) may be NULL. */
int tt_env_step_random(tt_env *env, uint64_t policy_seed, float *action_out, float *obs, float *reward,
                       uint8_t *done, const tt_info *info, int auto_reset, tt_stream_t stream);

/* Checkpoint / resume of the env batch (the reference only checkpoints networks, trainv2.py:210-244; SURVEY §8f-3
 * asks for env state too).  The blob is opaque device memory of tt_env_state_bytes(env) bytes holding every
 * per-env quantity (kinematic state, reward carry, counters, poses, goals, episode numbers); meta[4] carries the
 * handle's host-side mode {per-env-goal flag, reset seed, n_envs, version}.  Import requires the same n_envs. */
size_t tt_env_state_bytes(const tt_env *env);
int tt_env_export(tt_env *env, void *blob, uint64_t meta[4], tt_stream_t stream);
int tt_env_import(tt_env *env, const void *blob, const uint64_t meta[4], tt_stream_t stream);

/* K vector steps of the random policy in ONE launch (SURVEY.md §8d iii): each env stays in registers for
 * k_steps steps with in-kernel auto-reset; only the last observation is stored.  obs_out [N,23], reward_sum [N]
 * f32 (sum of the k_steps rewards) and episodes_done [N] i32 may each be NULL. */
int tt_env_rollout_random(tt_env *env, int k_steps, uint64_t policy_seed, float *obs_out, float *reward_sum,
                          int32_t *episodes_done, tt_stream_t stream);

/* Measurement hook (no reference counterpart): time each of the next `max_launches` step-kernel dispatches
 * with a HIP event pair bound to the dispatch itself (hipExtLaunchKernelGGL), on the stream they are launched
 * on; max_launches = 0 switches it off.  tt_env_profile_read waits for the recorded launches, adds them to the
 * running totals and returns sum of kernel durations [ms] and their count.  Not capturable into a hipGraph. */
int tt_env_profile(tt_env *env, int max_launches);
int tt_env_profile_read(tt_env *env, double *total_ms, int64_t *launches);

/* "random policy" of BASELINE.json config 2 as a stand-alone action generator: out[i] = U(-1,1) * pi/4 from Philox(seed, step). */
int tt_random_actions(int n, uint64_t seed, uint64_t step, float *out, tt_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused inference of the reference's networks (shapes of trainv2.py:404-407: 23 -> 400 -> 300 -> 1, LayerNorm
 * eps 1e-5) at f32 accuracy.  Pointers are torch parameter storages (row-major [out,in]):
 *   w1 [400,23] b1 g1 be1 [400] = fc1, bn1;  w2 [300,400] b2 g2 be2 [300] = fc2, bn2;  w3 [300] b3 [1] = mu / q;
 *   wa [300] ba [300] = action_value (critic only).  Other shapes return TT_EINVAL (callers fall back to torch).
 * Two kernels serve a forward: the exact-f32 MFMA kernel (csrc/ttnet.hip), and for n >= 1024 rows the split-f16
 * kernel (csrc/ttnet_split.hip: every f32 operand is, to within its own rounding, the sum of two round-to-nearest f16
 * pieces; three f16 MFMAs with f32 accumulation per product block) when split_ws is set: a caller-owned
 * device workspace of tt_mlp_split_ws_bytes() bytes, one per network and per stream that may run it concurrently,
 * into which each call re-packs fc2 before it runs (never stale; the contents are private to the library).
 * split_ws = NULL always selects the exact-f32 kernel.  Structs of gradients (tt_mlp_backward) ignore it. */
typedef struct tt_mlp_weights {
    const float *w1, *b1, *g1, *be1, *w2, *b2, *g2, *be2, *w3, *b3, *wa, *ba;
    int32_t in_dim, fc1_dims, fc2_dims;
    int32_t capped_grids;   /* with max_workgroups > 0: > 0 = only the first capped_grids grids are capped, the remaining tiles go
                               out in ONE grid over the whole chip (large n: the launches on other streams that the cap makes room
                               for are over after a few grids); 0 = every grid is capped */
    void *split_ws;
    int32_t ws_packed;      /* != 0: split_ws already holds the image of these weights (tt_mlp_split_pack): forwards do not
                               re-pack and read nothing but the image, so the weights may be updated beside them */
    int32_t max_workgroups; /* > 0: the split kernel's workgroups (one per CU is resident) go out in consecutive grids of at most
                               this many, leaving the other CUs to launches on other streams; 0 = one grid */
    void *split_ws_alt;     /* optional second image (ring addressing only): tt_mlp_split_pack[_and_sample] with a cursor then writes the
                               image of the parity of the step it opens (split_ws: even steps, split_ws_alt: odd steps) and
                               tt_actor_act_ring reads the one of the running step's parity -- so the opening launch of step
                               t+1 may run beside the policy launches of step t */
    void *fc2_img;          /* learn() kernels (tt_mlp_forward_save / _multi, tt_mlp_backward*): NULL = fc2 products on the exact-f32
                               MFMA straight from w2; else a caller-owned device buffer of tt_mlp_fc2_image_bytes() bytes (zeroed
                               once, then tt_mlp_fc2_image_pack) that holds w2 as pre-split f16 pieces in both orientations, and
                               the products run on the f16 MFMA at f32 accuracy (three per block, as in the split-f16 forward).
                               The CALLER keeps it equal to w2: optimizer steps through tt_mlp_backward_weights /
                               tt_adam_soft_update with `images` rewrite every element they update; after any other change of
                               w2 call tt_mlp_fc2_image_pack again. */
} tt_mlp_weights;
/* fc2 as pre-split f16 pieces (x64, h = rn16, m = rn16 of the remainder) in MFMA-fragment order (1 KB = 64 lanes x 8 halves =
 * one coalesced 16-byte load per lane): a forward half [20 neuron tiles][13 k32 steps] and a backward half (dH1 = dX2 * W2)
 * [28 column tiles][10 k32 steps], an h and an m plane of each; zero padding (layout private to csrc/ttlearn.hip). */
uint64_t tt_mlp_fc2_image_bytes(void);
int tt_mlp_fc2_image_pack(const tt_mlp_weights *w, tt_stream_t stream);   /* w->fc2_img <- pieces of w->w2 */
/* images an optimizer step keeps current (either may be NULL): the updated network's and its target's fc2_img */
typedef struct tt_fc2_images {
    void *net, *target;
} tt_fc2_images;
uint64_t tt_mlp_split_ws_bytes(void);
/* Write the split kernel's image of `w` (fc2 and fc1 as pre-split f16 fragments, per-neuron vectors, head bias) into ws
 * (tt_mlp_split_ws_bytes() bytes).  bump (may be NULL): a device int64 that this launch increments by one -- the step
 * counter of a pipelined loop whose learn() chain ends with this pack. */
int tt_mlp_split_pack(const tt_mlp_weights *w, int critic, void *ws, int64_t *bump, const tt_ring_cursor *cursor,
                      tt_stream_t stream);   /* cursor (may be NULL): also writes the ring cursor of the step this launch opens */

/* ActorNetwork.forward (DDPG/networks.py:138-147) for n rows: mu_out [n] = tanh(mu(...)). */
int tt_actor_forward(int n, const float *obs /*[n,23]*/, const tt_mlp_weights *w, float *mu_out, tt_stream_t stream);

/* Agent.choose_action for n envs (DDPG_agent.py:36-49) + OUActionNoise.__call__ (noise.py:13-17) + the caller's
 * scaling (trainv2.py:516) in one launch: ou_state [n] is advanced in place (x <- x*(1-theta_dt) + sigma_sqrt_dt*N(0,1),
 * N from Philox(seed, env, step [+ *step_dev]) + Box-Muller; restarted at 0 where done_prev[i] != 0, trainv2.py:492),
 * act_raw_out [n] = mu + x (the action the replay stores, trainv2.py:525), act_scaled_out [n] = clip(.,-1,1)*high
 * (what env.step gets).  mu_out and done_prev may be NULL.  step_dev (device int64) may be NULL. */
int tt_actor_act(int n, const float *obs, const tt_mlp_weights *w, float *ou_state, const uint8_t *done_prev,
                 uint64_t seed, uint64_t step, const int64_t *step_dev, float theta_dt, float sigma_sqrt_dt, float high,
                 float *mu_out, float *act_raw_out, float *act_scaled_out, tt_stream_t stream);

/* tt_actor_act on ring slot t (observations in, stored actions out, done flags of slot t-1 restart the noise): `w` must carry a
 * caller-kept image (ws_packed).  step_dev (required) is the ring's step counter -- the one tt_env_step_ring's launch
 * advances: besides keying the noise, its parity picks the cursor pair and (split_ws_alt) the image of the running step, and
 * the launch copies that cursor to cursor[0..3] for the env step that follows it on the same stream. */
int tt_actor_act_ring(int n, const tt_ring_view *ring, const tt_mlp_weights *w, float *ou_state, uint64_t seed, uint64_t step,
                      const int64_t *step_dev, float theta_dt, float sigma_sqrt_dt, float high, float *act_scaled_out,
                      tt_stream_t stream);

/* ReplayBuffer.sample_buffer (DDPG/replay_buffer.py:23-34: uniform WITH replacement) on the device trajectory ring
 * obs [slots,N,23] f32, act/rew [slots,N] f32, done [slots,N] u8 (transition (t,e) = obs[t][e], act[t][e], rew[t][e],
 * obs[t+1][e], done[t][e]); *k_dev = vector steps completed, read on the device, so a captured hipGraph draws new
 * indices at every replay.  Outputs: s_out, s2_out [batch,23], a_out, r_out [batch] f32, d_out [batch] u8,
 * idx_out [batch,2] i32 (slot, env) or NULL.
 * reserve: 0, or the number of most recent ring slots a CONCURRENT env step may be writing (a pipelined loop samples
 * beside the step launch: then *k_dev counts the steps completed before that launch and reserve = 1 keeps the draw off the
 * observation row it overwrites; two env steps may be under way beside a draw that runs ahead of them: reserve = 2); the window is min(*k_dev - lag, slots - 1 - reserve) steps ending lag steps before *k_dev
 * (lag: steps counted by *k_dev that may still be under way on another stream -- a loop whose learn() chain runs ahead of its
 * env steps: DDPGRollout's pipelined order uses lag = 1, reserve = 2).
 * side (may be NULL): stand-alone transitions that are not part of any env's trajectory -- the expert tuples
 * `(obs, action / radians(45), reward, obs_next, done)` that trainv2.py:457-466 re-inserts with agent.remember
 * (produced by exp_gen.py:77-110).  They take part in the same uniform draw: with `count` side transitions and R
 * intact ring transitions every one of the count + R has probability 1/(count + R); a side draw reports
 * idx_out = (-1, j). */
typedef struct tt_side_buffer {
    const float *obs, *act, *rew, *obs2; /* [count,23], [count], [count], [count,23] f32 */
    const uint8_t *done;                 /* [count] */
    int32_t count, reserved_;
} tt_side_buffer;
int tt_ring_sample(int batch, int n_envs, int slots, const int64_t *k_dev, const float *obs, const float *act,
                   const float *rew, const uint8_t *done, uint64_t seed, int reserve, int lag, const tt_side_buffer *side,
                   float *s_out, float *a_out, float *r_out, float *s2_out, uint8_t *d_out, int32_t *idx_out,
                   tt_stream_t stream);

/* tt_mlp_split_pack + tt_ring_sample in ONE launch: what opens a vector step of a pipelined loop (the policy's image from
 * the actor's current weights; the first batch of the step's learn()).  The members of tt_sample_args are tt_ring_sample's
 * arguments. */
typedef struct tt_sample_args {
    int32_t batch, n_envs, slots, reserve;      /* (lag: last member) */
    const int64_t *k_dev;
    const float *obs, *act, *rew;
    const uint8_t *done;
    uint64_t seed;
    const tt_side_buffer *side;
    float *s_out, *a_out, *r_out, *s2_out;
    uint8_t *d_out;
    int32_t *idx_out;
    int32_t lag;             /* the newest `lag` steps counted by *k_dev may still be under way (tt_ring_sample) */
    int32_t draws;           /* tt_mlp_split_pack_and_sample: 0 or 1 = one draw; d > 1 = d draws of `batch` rows each from the SAME
                              * window in the one launch -- draw u uses seed + u * seed_stride and fills rows [u * batch, (u + 1) *
                              * batch) of the output buffers (sized for d * batch rows): the batches of all the learn() calls of one
                              * vector step (each of them exactly what its own tt_ring_sample with that seed would draw), so that
                              * none of them has the ring's latency on its chain.  Other entry points take one draw and ignore it */
    uint64_t seed_stride;
    const int32_t *step_progress; /* tt_mlp_forward_multi_sampled only; NULL or the ring's cursor + TT_CURSOR_PROGRESS: the launch then first
                              * waits until *step_progress >= *k_dev -- the env step whose transitions close the draw's window is over
                              * (k_dev must then be a counter that runs `lag` ahead of the steps completed, as a pipelined loop's does);
                              * bounded like the image hand-over, same give-up word (cursor[TT_CURSOR_GAVE_UP]) */
} tt_sample_args;
int tt_mlp_split_pack_and_sample(const tt_mlp_weights *w, int critic, void *ws, const tt_sample_args *sample,
                                 const tt_ring_cursor *cursor, tt_stream_t stream);

/* CriticNetwork.forward (DDPG/networks.py:55-68) for n rows: q_out [n]. */
int tt_critic_forward(int n, const float *obs /*[n,23]*/, const float *action /*[n]*/, const tt_mlp_weights *w,
                      float *q_out, tt_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Hand-fused learn() (csrc/ttlearn.hip) for the same network shapes: Agent.learn (DDPG/DDPG_agent.py:72-106) as a
 * dozen launches.  All buffers are caller-owned device memory, row-major f32. */
typedef struct tt_mlp_saved {   /* what a forward keeps for its backward */
    float *xh1, *h1;            /* [B,400] LayerNorm1-normalised fc1 output (before gamma/beta); post-ReLU activation */
    float *xh2, *h2;            /* [B,300] same for fc2 (critic: h2 after adding action_value(a)) */
    float *rstd1, *rstd2;       /* [B] 1/sqrt(var + eps) of the two LayerNorms */
} tt_mlp_saved;

/* Forward of the actor (critic = 0: out = tanh(mu(.))) or the critic (critic = 1: out = Q(s,a)) on a small batch,
 * 16 rows per workgroup with the columns split over its 4 waves.  saved may be NULL (inference only); dq_da [B]
 * (critic only, may be NULL) receives dQ/da. */
int tt_mlp_forward_save(int n, int critic, const float *obs, const float *action, const tt_mlp_weights *w, float *out,
                        const tt_mlp_saved *saved, float *dq_da, tt_stream_t stream);

/* Up to four tt_mlp_forward_save / tt_critic_state_forward jobs on n rows each in ONE launch (learn()'s first phase:
 * target actor on s', target critic's state branch on s', Q(s,a), mu(s)).  critic = 0: actor (action, dq_da, z_state
 * ignored); critic = 1 with z_state set: state branch only (action and out may be NULL); saved / dq_da may be NULL. */
typedef struct tt_fwd_job {
    int32_t critic, reserved_;
    const float *obs, *action;
    const tt_mlp_weights *w;
    float *out;
    const tt_mlp_saved *saved;
    float *dq_da, *z_state;
} tt_fwd_job;
int tt_mlp_forward_multi(int n, int count, const tt_fwd_job *jobs, tt_stream_t stream);
/* The same with the replay draw of ReplayBuffer.sample_buffer (DDPG/replay_buffer.py:23-34) made BY this launch: `sample`
 * are tt_ring_sample's arguments (batch = n); every job's obs must be sample->s_out or sample->s2_out and a critic job's
 * action sample->a_out -- the workgroups read those rows straight from the ring, and the five batch buffers are filled on
 * the way for the launches that follow (at least one job on s and one on s').  One launch and one dependent launch boundary
 * less per learn() than tt_ring_sample followed by tt_mlp_forward_multi; the same draw, bit for bit. */
/* k_snapshot (may be NULL): *sample->k_dev as this launch saw it, left for a LATER launch that must use the same step number
 * although the counter moves in between (tt_image_job below). */
int tt_mlp_forward_multi_sampled(int n, int count, const tt_fwd_job *jobs, const tt_sample_args *sample, int64_t *k_snapshot,
                                 tt_stream_t stream);

/* The target critic in two pieces, so that its state branch can run NEXT TO the target actor that produces its action:
 * tt_critic_state_forward: z_state [n,300] = bn2(fc2(relu(bn1(fc1(s))))) (networks.py:55-61, before the action enters);
 * tt_critic_head_td: q' = q(relu(z_state + action_value(a))) (networks.py:62-68) and the TD target
 * y = r + gamma * q' * (1 - done) (DDPG_agent.py:89-93) in one launch; also advances *step_dev (may be NULL) like
 * tt_td_target; q_out [n] may be NULL. */
int tt_critic_state_forward(int n, const float *obs, const tt_mlp_weights *w, float *z_state, tt_stream_t stream);
int tt_critic_head_td(int n, const float *z_state, const float *action, const tt_mlp_weights *w, const float *reward,
                      const uint8_t *done, float gamma, float *y, float *q_out, int64_t *step_dev, tt_stream_t stream);

/* Backward of one net on the batch (autograd of networks.py:55-68 / 138-147): workspace ws holds the per-row
 * gradients (dpre [B], dz, dx2 [B,300], dy1, dx1 [B,400]); grads has the layout of tt_mlp_weights and receives
 * d(loss)/d(parameter) for every parameter (overwritten, not accumulated).
 *   mode 0: d_out [B] = d(loss)/d(out) given;
 *   mode 1: d(loss)/d(out) = scale*(out - y)   (critic: loss = mse_loss(y, q), scale = 2/B; DDPG_agent.py:96-97);
 *   mode 2: d(loss)/d(out) = scale*aux         (actor: loss = -mean Q(s, mu(s)), aux = dQ/da from
 *                                               tt_mlp_forward_save on the critic, scale = -1/B; DDPG_agent.py:101-103). */
typedef struct tt_mlp_bwd_ws {
    float *dpre, *dz, *dx2, *dy1, *dx1;
} tt_mlp_bwd_ws;
/* td (optional, critic with mode 1 only; y may then be NULL): the work of tt_critic_head_td done as the prologue of the
 * critic's backward launch instead of a launch of its own -- q' from the target critic's state branch z_state [n,300] and
 * the target actor's action mu_target [n], y_out [n] = r + gamma q' (1 - done) (used as y and written out), q_out [n] or
 * NULL, *step_dev advanced by 1 (may be NULL). */
typedef struct tt_td_input {
    const float *z_state, *mu_target;
    const tt_mlp_weights *target_critic;
    const float *reward;
    const uint8_t *done;
    float gamma, reserved_;
    float *y_out, *q_out;
    int64_t *step_dev;
    int64_t *window_dev;   /* optional: a second device counter advanced by 1 -- the sampling-window counter of a pipelined
                              loop (tt_ring_sample's k_dev), moved on by the last learn() of a vector step */
    /* optional (tt_mlp_backward_rows_pair only): the launch that advances *step_dev also leaves Adam's bias corrections of
     * the NEW step t for the pair (adam_beta1, adam_beta2) in bias_corr_out[0..4] = {t (int32 bits), beta1, beta2,
     * 1 - beta1^t, 1 - beta2^t}, evaluated once in f64 on a workgroup of its own; the optimizer launches given the same
     * buffer (tt_mlp_backward_weights / tt_adam_soft_update: bias_corr) use it when step and betas match and otherwise
     * evaluate the two pow() themselves, in every thread: ~1.1 us per launch (DDPG/networks.py:49-50,133: torch.optim.Adam) */
    float *bias_corr_out;
    float adam_beta1, adam_beta2;
} tt_td_input;
#define TT_BIAS_CORR_FLOATS 8
int tt_mlp_backward(int n, int critic, int mode, float scale, const float *obs, const float *action, const float *d_out,
                    const float *out, const float *y, const float *aux, const tt_mlp_weights *w,
                    const tt_mlp_saved *saved, const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, const tt_td_input *td,
                    tt_stream_t stream);

/* tt_mlp_backward with the optimizer step of tt_adam_soft_update applied in the weight-gradient launch itself (each
 * gradient element is finished by exactly one workgroup, which then updates that parameter, its Adam moments and its
 * target): one launch less per network.  count = 10 (actor) / 12 (critic) tensors in tt_mlp_weights order; the arrays
 * are HOST arrays of device pointers; grads still receives the gradients.  For the single-rank path: with data-parallel
 * ranks the gradients are all-reduced between tt_mlp_backward and tt_adam_soft_update instead. */
int tt_mlp_backward_adam(int n, int critic, int mode, float scale, const float *obs, const float *action, const float *d_out,
                         const float *out, const float *y, const float *aux, const tt_mlp_weights *w,
                         const tt_mlp_saved *saved, const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, int count,
                         float *const *params, float *const *exp_avg, float *const *exp_avg_sq, float *const *targets,
                         const int64_t *step_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                         float tau, const tt_td_input *td, tt_stream_t stream);

/* learn()'s second phase in the form the rollout loop uses.  The actor's per-row backward is linear in the row's
 * d(loss)/d(pre-tanh) = -(1/B) dQ/da (1 - mu^2), and only dQ/da needs the UPDATED critic (DDPG_agent.py:100-103): so
 *   tt_mlp_backward_rows_pair   runs the critic's per-row backward (mode 1, TD target in its prologue: tt_td_input) and, on
 *                               other workgroups of the SAME launch, the actor's per-row backward for a unit gradient
 *                               (ws_actor receives d(.)/d(pre-tanh) = 1 per row; ws_critic != ws_actor);
 *   tt_mlp_backward_weights     is the weight-gradient launch alone (what tt_mlp_backward[_adam] runs second): gradients of
 *                               every parameter from saved + ws, with row b of ws counted row_scale * row_dq_da[b] *
 *                               (1 - row_mu[b]^2) times when row_dq_da / row_mu are given (both or neither), and with
 *                               Adam + soft update in the same launch when count != 0 (arguments as tt_mlp_backward_adam).
 * Sequence of a learn(): tt_mlp_forward_multi, tt_mlp_backward_rows_pair, tt_mlp_backward_weights(critic), tt_mlp_forward_save
 * (critic on (s, mu(s)) with dq_da), tt_mlp_backward_weights(actor, row_dq_da = dq_da, row_mu = mu, row_scale = -1/B). */
/* image (may be NULL): the pack of a vector step's policy image (tt_mlp_split_pack with a cursor: image of the step's parity,
 * ring cursor, image epoch) carried by THIS launch on workgroups of its own -- the actor's weights are not written before
 * the launch after the next, the pack needs ~5 us of the launch's ~14, and a loop's learn chain is one launch and one dependent
 * boundary shorter than with an opening pack launch.  image->cursor->k_dev must be a word no launch writes meanwhile
 * (tt_mlp_forward_multi_sampled: k_snapshot): this launch also advances the step / window counters. */
typedef struct tt_image_job {
    const tt_mlp_weights *actor;        /* its split_ws / split_ws_alt are the image buffers (as for tt_mlp_split_pack) */
    const tt_ring_cursor *cursor;
} tt_image_job;
int tt_mlp_backward_rows_pair(int n, float scale_critic, const float *q_out, const tt_mlp_weights *critic,
                              const tt_mlp_saved *saved_critic, const tt_mlp_bwd_ws *ws_critic, const tt_td_input *td,
                              const float *mu_out, const tt_mlp_weights *actor, const tt_mlp_saved *saved_actor,
                              const tt_mlp_bwd_ws *ws_actor, const tt_image_job *image, tt_stream_t stream);
int tt_mlp_backward_weights(int n, int critic, const float *obs, const float *action, const tt_mlp_saved *saved,
                            const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, const float *row_dq_da, const float *row_mu,
                            float row_scale, int count, float *const *params, float *const *exp_avg, float *const *exp_avg_sq,
                            float *const *targets, const int64_t *step_dev, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float tau, const tt_fc2_images *images /* may be NULL */,
                            const float *bias_corr /* tt_td_input.bias_corr_out or NULL */, tt_stream_t stream);

/* The last two launches of the single-rank sequence above in ONE grid: tt_mlp_forward_save(critic on (s, mu), dq_da) and
 * tt_mlp_backward_weights(actor, row_dq_da = dq_da, row_mu = mu, row_scale, Adam + soft update + images).  The weight-gradient
 * workgroups request everything else they read, then wait IN DEVICE MEMORY for the row workgroups' dQ/da instead of behind a launch
 * boundary.  tail_words: 64 + 2 n device ints, set to -1 by the caller at creation and whenever it sets *step_dev back: [0, 64) one
 * hint word per row workgroup, [64, 64 + 2 n) per ROW one 8-byte word {learn step *step_dev, float bits of dQ/da} written by ONE
 * agent-scope atomic store -- a reader that finds the step it waits for holds the value of that step, so the hand-over needs no
 * ordering between two locations and no cache maintenance.  The wait is bounded (0.25 s): a thread that gives up stores the step
 * into *gave_up_host (one int of device-visible host memory, may be NULL) and goes on -- the caller must treat that learn() as
 * failed.  Same results as the two launches, bit for bit (dq_da [n] is written as well).  n <= 1024 rows. */
int tt_mlp_actor_tail(int n, const float *obs, const float *mu, const tt_mlp_weights *critic, float *q_out, float *dq_da,
                      const tt_mlp_saved *saved, const tt_mlp_bwd_ws *ws, const tt_mlp_weights *grads, float row_scale, int count,
                      float *const *params, float *const *exp_avg, float *const *exp_avg_sq, float *const *targets,
                      const int64_t *step_dev, float lr, float beta1, float beta2, float eps, float weight_decay, float tau,
                      const tt_fc2_images *images, const float *bias_corr, int32_t *tail_words, int32_t *gave_up_host,
                      tt_stream_t stream);

/* optimizer.step() of torch.optim.Adam (weight decay folded into the gradient; networks.py:49-50,133) for `count`
 * (<= 12) parameter tensors in one launch, then the soft update of the matching target tensors
 * (Agent.update_network_parameters, DDPG_agent.py:108-131; targets NULL = none).  The arrays are HOST arrays of device
 * pointers; *step_dev (device) is the 1-based count of this step. */
int tt_adam_soft_update(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                        float *const *exp_avg_sq, float *const *targets, const int32_t *numel, const int64_t *step_dev,
                        float lr, float beta1, float beta2, float eps, float weight_decay, float tau,
                        const tt_fc2_images *images /* may be NULL; tensors in tt_mlp_weights order (w2 = index 4) */,
                        const float *bias_corr /* tt_td_input.bias_corr_out or NULL */, tt_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Peer-to-peer gradient exchange of data-parallel ranks (one process per GPU of one node): the mean over the ranks of the
 * critic's / the actor's gradient at the reference's two optimizer sites (DDPG/DDPG_agent.py:95-104) WITHOUT a collective
 * launch on learn()'s chain.  Every rank owns two blocks of device memory -- per site a flat f32 gradient buffer
 * (tt_mlp_weights order: what tt_mlp_backward_weights writes when `grads` points into it), and, in fine-grained memory, one
 * arrival word per site and rank -- that its peers open through hipIpcMemHandles; tt_adam_soft_update_p2p is
 * tt_adam_soft_update whose gradient is
 *      g[i] = (G_0[i] + G_1[i] + ... + G_{world-1}[i]) / world        (summed in rank order: the same bits on every rank)
 * read straight from the ranks' buffers (xGMI loads at system scope).  Hand-over, per site and learn step t = *step_dev:
 * the first workgroup of rank r's launch -- which starts only when the launch that wrote G_r (same stream) is complete and
 * written back -- stores t into word [site][r] of EVERY rank's block (system scope, after a system-scope release); every
 * workgroup waits until its own block holds t in the words of all ranks, then acquires.  A rank overwrites G_r[site] only
 * in its next backward launch of that site, which lies behind its update of the OTHER site, whose wait saw every peer past
 * its reads of this one: two sites used in alternation (critic, actor, critic, ...) need no second barrier.  The wait is
 * bounded (tt_p2p_set_timeout, default 2 s): a launch that gives up marks a host-visible word (tt_p2p_gave_up) and goes on
 * with whatever the buffers hold -- the caller must treat the ranks as diverged.  Nothing here needs a process group; the
 * caller moves the TT_P2P_HANDLE_BYTES of tt_p2p_export between the processes (e.g. torch.distributed.all_gather_object on any backend). */
typedef struct tt_p2p tt_p2p;
#define TT_P2P_MAX_RANKS 8
#define TT_P2P_MAX_SITES 4
#define TT_P2P_HANDLE_BYTES 128      /* two hipIpcMemHandle_t: the flag block, the gradient block */
int tt_p2p_create(int device, int rank, int world, int sites, const int32_t *numel /*[sites] floats per site*/, tt_p2p **out);
int tt_p2p_destroy(tt_p2p *x);                               /* closes the peers' blocks, frees its own (peers must be done with it) */
int tt_p2p_export(const tt_p2p *x, void *handle_out /*[TT_P2P_HANDLE_BYTES]*/);
int tt_p2p_attach(tt_p2p *x, int peer, const void *handle /*[TT_P2P_HANDLE_BYTES] of rank `peer`*/);
float *tt_p2p_grad(const tt_p2p *x, int site);                /* this rank's gradient buffer of `site` (device memory), or NULL */
int tt_p2p_reset(tt_p2p *x, tt_stream_t stream);              /* own arrival words back to 0 (a step counter set back: resume);
                                                                 the caller keeps every rank out of the exchange meanwhile */
int tt_p2p_set_timeout(tt_p2p *x, double seconds);
int tt_p2p_gave_up(const tt_p2p *x);                          /* 0, or the step whose wait was abandoned; reads host memory only */
const char *tt_p2p_last_error(const tt_p2p *x);
/* tt_adam_soft_update with the gradient taken from the exchange (above).  numel[0..count) must add up to the site's size;
 * every rank of the exchange must launch it for the same site with the same *step_dev. */
int tt_adam_soft_update_p2p(tt_p2p *x, int site, int count, float *const *params, float *const *exp_avg,
                            float *const *exp_avg_sq, float *const *targets, const int32_t *numel, const int64_t *step_dev,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float tau,
                            const tt_fc2_images *images, const float *bias_corr, tt_stream_t stream);

/* target = rewards + gamma * critic_value_ with critic_value_[done] = 0 (DDPG_agent.py:89-93); also advances the
 * learn-step counter *step_dev (may be NULL) that tt_adam_soft_update reads. */
int tt_td_target(int n, const float *reward, const float *q_next, const uint8_t *done, float gamma, float *y,
                 int64_t *step_dev, tt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TTENV_H */
