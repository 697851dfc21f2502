"""ctypes binding of libttenv.so (include/ttenv.h).  The product has no CPU fallback: if the
library is missing or a call fails, this raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TT_LIB_PATH") or os.path.join(HERE, "libttenv.so")  # override: kernel-variant A/B runs

OBS_DIM = 23
MAX_EPISODE_STEPS = 4095      # TT_MAX_EPISODE_STEPS
TT_OK, TT_EINVAL, TT_ENOMEM, TT_EHIP, TT_ENODEV = 0, -1, -2, -3, -4
F_JACKKNIFE, F_OUT_OF_MAP, F_MAX_STEPS, F_GOAL_REACHED, F_GOAL_PASSED, F_EXCESSIVE_BACK, F_SUCCESS = (1 << i for i in range(7))
VIOLATIONS = ("none", "jackknife", "jackknife_warning", "major_boundary", "minor_boundary", "past_the_goal",
              "max_step", "excessive_backward")
INFO_ROWS = ("total_reward", "progress_reward", "heading_reward", "orientation_reward", "staged_success",
             "safety_penalty", "exploration_bonus", "final_success_bonus", "backward_penalty", "smoothness_penalty",
             "cumulative_backward", "movement_budget")
NINFO = len(INFO_ROWS)


class TTParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("L1", "L2", "hitch_offset", "v1x", "dt", "map_min_x", "map_max_x", "map_min_y",
                                          "map_max_y", "max_steer", "position_threshold", "orientation_threshold",
                                          "step_length")] + \
               [("extra_steps", C.c_int32), ("fixed_max_steps", C.c_int32), ("term_mask", C.c_uint32),
                ("variant", C.c_int32), ("stateless_reward", C.c_int32), ("reserved_", C.c_int32),
                ("goal", C.c_double * 3), ("reset_lo", C.c_double * 3),
                ("reset_hi", C.c_double * 3)]


class TTInfo(C.Structure):
    _fields_ = [("comp", C.c_void_p), ("violation", C.c_void_p), ("flags", C.c_void_p)]


class TTMlpWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "w3", "b3", "wa", "ba")] + \
               [("in_dim", C.c_int32), ("fc1_dims", C.c_int32), ("fc2_dims", C.c_int32), ("capped_grids", C.c_int32),
                ("split_ws", C.c_void_p), ("ws_packed", C.c_int32), ("max_workgroups", C.c_int32), ("split_ws_alt", C.c_void_p), ("fc2_img", C.c_void_p)]


class TTFc2Images(C.Structure):
    _fields_ = [("net", C.c_void_p), ("target", C.c_void_p)]


class TTMlpSaved(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("xh1", "h1", "xh2", "h2", "rstd1", "rstd2")]


class TTFwdJob(C.Structure):
    _fields_ = [("critic", C.c_int32), ("reserved_", C.c_int32), ("obs", C.c_void_p), ("action", C.c_void_p),
                ("w", C.POINTER(TTMlpWeights)), ("out", C.c_void_p), ("saved", C.POINTER(TTMlpSaved)),
                ("dq_da", C.c_void_p), ("z_state", C.c_void_p)]


class TTTdInput(C.Structure):
    _fields_ = [("z_state", C.c_void_p), ("mu_target", C.c_void_p), ("target_critic", C.POINTER(TTMlpWeights)),
                ("reward", C.c_void_p), ("done", C.c_void_p), ("gamma", C.c_float), ("reserved_", C.c_float),
                ("y_out", C.c_void_p), ("q_out", C.c_void_p), ("step_dev", C.c_void_p), ("window_dev", C.c_void_p),
                ("bias_corr_out", C.c_void_p), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float)]


class TTSideBuffer(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p), ("obs2", C.c_void_p), ("done", C.c_void_p),
                ("count", C.c_int32), ("reserved_", C.c_int32)]


class TTRingView(C.Structure):
    _fields_ = [("cursor", C.c_void_p), ("obs", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p),
                ("n_envs", C.c_int32), ("slots", C.c_int32)]


class TTRingCursor(C.Structure):
    _fields_ = [("k_dev", C.c_void_p), ("slots", C.c_int32), ("reserved_", C.c_int32), ("cursor", C.c_void_p)]


class TTSampleArgs(C.Structure):
    _fields_ = [("batch", C.c_int32), ("n_envs", C.c_int32), ("slots", C.c_int32), ("reserve", C.c_int32), ("k_dev", C.c_void_p),
                ("obs", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p), ("seed", C.c_uint64),
                ("side", C.POINTER(TTSideBuffer)), ("s_out", C.c_void_p), ("a_out", C.c_void_p), ("r_out", C.c_void_p),
                ("s2_out", C.c_void_p), ("d_out", C.c_void_p), ("idx_out", C.c_void_p), ("lag", C.c_int32),
                ("draws", C.c_int32), ("seed_stride", C.c_uint64), ("step_progress", C.c_void_p)]


class TTImageJob(C.Structure):
    _fields_ = [("actor", C.POINTER(TTMlpWeights)), ("cursor", C.POINTER(TTRingCursor))]


class TTMlpBwdWs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dpre", "dz", "dx2", "dy1", "dx1")]


class TTError(RuntimeError):
    pass


_P, _I, _U64 = C.c_void_p, C.c_int, C.c_uint64
_SIGNATURES = {
    "tt_version": (C.c_int, []),
    "tt_last_error": (C.c_char_p, [_P]),
    "tt_params_default": (C.c_int, [_I, C.POINTER(TTParams)]),
    "tt_env_create": (C.c_int, [_I, _I, C.POINTER(TTParams), C.POINTER(_P)]),
    "tt_env_destroy": (C.c_int, [_P]),
    "tt_env_num_envs": (C.c_int, [_P]),
    "tt_env_reset": (C.c_int, [_P, _P, _U64, _P, _P]),
    "tt_env_set_reset_pool": (C.c_int, [_P, _P, _I]),
    "tt_env_set_pose": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _P]),
    "tt_env_set_attrs": (C.c_int, [_P, _P, _I, _P, _P, _P, _P]),
    "tt_env_set_state": (C.c_int, [_P, _P, _I, _P, _P]),
    "tt_env_get_state": (C.c_int, [_P, _P, _P]),
    "tt_env_set_max_steps": (C.c_int, [_P, _P, _I, _P, _P]),
    "tt_env_set_steps": (C.c_int, [_P, _P, _I, _P, _P]),
    "tt_env_get_episode": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "tt_env_observe": (C.c_int, [_P, _P, _P, _P]),
    "tt_env_set_step_counter": (C.c_int, [_P, _P]),
    "tt_env_step": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(TTInfo), _I, _P]),
    "tt_env_step_random": (C.c_int, [_P, _U64, _P, _P, _P, _P, C.POINTER(TTInfo), _I, _P]),
    "tt_env_state_bytes": (C.c_size_t, [_P]),
    "tt_env_export": (C.c_int, [_P, _P, C.POINTER(C.c_uint64 * 4), _P]),
    "tt_env_import": (C.c_int, [_P, _P, C.POINTER(C.c_uint64 * 4), _P]),
    "tt_env_rollout_random": (C.c_int, [_P, _I, _U64, _P, _P, _P, _P]),
    "tt_env_profile": (C.c_int, [_P, _I]),
    "tt_env_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "tt_mlp_split_ws_bytes": (C.c_uint64, []),
    "tt_mlp_split_pack": (C.c_int, [C.POINTER(TTMlpWeights), _I, _P, _P, C.POINTER(TTRingCursor), _P]),
    "tt_mlp_split_pack_and_sample": (C.c_int, [C.POINTER(TTMlpWeights), _I, _P, C.POINTER(TTSampleArgs), C.POINTER(TTRingCursor), _P]),
    "tt_actor_act_ring": (C.c_int, [_I, C.POINTER(TTRingView), C.POINTER(TTMlpWeights), _P, _U64, _U64, _P, C.c_float, C.c_float,
                                    C.c_float, _P, _P]),
    "tt_env_step_ring": (C.c_int, [_P, _P, C.POINTER(TTRingView), _I, _P]),
    "tt_actor_forward": (C.c_int, [_I, _P, C.POINTER(TTMlpWeights), _P, _P]),
    "tt_actor_act": (C.c_int, [_I, _P, C.POINTER(TTMlpWeights), _P, _P, _U64, _U64, _P, C.c_float, C.c_float, C.c_float,
                               _P, _P, _P, _P]),
    "tt_ring_sample": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _U64, _I, _I, C.POINTER(TTSideBuffer), _P, _P, _P, _P, _P, _P, _P]),
    "tt_critic_forward": (C.c_int, [_I, _P, _P, C.POINTER(TTMlpWeights), _P, _P]),
    "tt_mlp_forward_save": (C.c_int, [_I, _I, _P, _P, C.POINTER(TTMlpWeights), _P, C.POINTER(TTMlpSaved), _P, _P]),
    "tt_mlp_forward_multi": (C.c_int, [_I, _I, C.POINTER(TTFwdJob), _P]),
    "tt_mlp_forward_multi_sampled": (C.c_int, [_I, _I, C.POINTER(TTFwdJob), C.POINTER(TTSampleArgs), _P, _P]),
    "tt_critic_state_forward": (C.c_int, [_I, _P, C.POINTER(TTMlpWeights), _P, _P]),
    "tt_critic_head_td": (C.c_int, [_I, _P, _P, C.POINTER(TTMlpWeights), _P, _P, C.c_float, _P, _P, _P, _P]),
    "tt_mlp_backward": (C.c_int, [_I, _I, _I, C.c_float, _P, _P, _P, _P, _P, _P, C.POINTER(TTMlpWeights),
                                  C.POINTER(TTMlpSaved), C.POINTER(TTMlpBwdWs), C.POINTER(TTMlpWeights), C.POINTER(TTTdInput),
                                  _P]),
    "tt_mlp_backward_adam": (C.c_int, [_I, _I, _I, C.c_float, _P, _P, _P, _P, _P, _P, C.POINTER(TTMlpWeights),
                                       C.POINTER(TTMlpSaved), C.POINTER(TTMlpBwdWs), C.POINTER(TTMlpWeights), _I, _P, _P, _P,
                                       _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.POINTER(TTTdInput), _P]),
    "tt_mlp_backward_rows_pair": (C.c_int, [_I, C.c_float, _P, C.POINTER(TTMlpWeights), C.POINTER(TTMlpSaved), C.POINTER(TTMlpBwdWs),
                                            C.POINTER(TTTdInput), _P, C.POINTER(TTMlpWeights), C.POINTER(TTMlpSaved),
                                            C.POINTER(TTMlpBwdWs), C.POINTER(TTImageJob), _P]),
    "tt_mlp_backward_weights": (C.c_int, [_I, _I, _P, _P, C.POINTER(TTMlpSaved), C.POINTER(TTMlpBwdWs), C.POINTER(TTMlpWeights),
                                          _P, _P, C.c_float, _I, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.POINTER(TTFc2Images), _P, _P]),
    "tt_mlp_actor_tail": (C.c_int, [_I, _P, _P, C.POINTER(TTMlpWeights), _P, _P, C.POINTER(TTMlpSaved), C.POINTER(TTMlpBwdWs),
                                    C.POINTER(TTMlpWeights), C.c_float, _I, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.POINTER(TTFc2Images), _P, _P, _P, _P]),
    "tt_adam_soft_update": (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.POINTER(TTFc2Images), _P, _P]),
    "tt_p2p_create": (C.c_int, [_I, _I, _I, _I, _P, C.POINTER(_P)]),
    "tt_p2p_destroy": (C.c_int, [_P]),
    "tt_p2p_export": (C.c_int, [_P, _P]),
    "tt_p2p_attach": (C.c_int, [_P, _I, _P]),
    "tt_p2p_grad": (_P, [_P, _I]),
    "tt_p2p_reset": (C.c_int, [_P, _P]),
    "tt_p2p_set_timeout": (C.c_int, [_P, C.c_double]),
    "tt_p2p_gave_up": (C.c_int, [_P]),
    "tt_p2p_last_error": (C.c_char_p, [_P]),
    "tt_adam_soft_update_p2p": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.POINTER(TTFc2Images), _P, _P]),
    "tt_mlp_fc2_image_bytes": (C.c_uint64, []),
    "tt_mlp_fc2_image_pack": (C.c_int, [C.POINTER(TTMlpWeights), _P]),
    "tt_td_target": (C.c_int, [_I, _P, _P, _P, C.c_float, _P, _P, _P]),
    "tt_random_actions": (C.c_int, [_I, _U64, _U64, _P, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """dlopen libttenv.so and declare every entry point of include/ttenv.h."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch first: it brings its own copy of the HIP runtime (torch/lib/libamdhip64.so, loaded by path), and libttenv.so then
        # binds to that copy by its soname.  The other way round the process holds TWO runtimes -- /opt/rocm's for this library,
        # torch's for the tensors it is handed -- and tt_env_create finds no device in its own (seen on the GPU box with
        # `g.build(); g.smoke()`, where build() loaded the library before anything had imported torch)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, handle=None):
    if rc != TT_OK:
        msg = load().tt_last_error(handle)
        raise TTError(f"libttenv error {rc}: {msg.decode() if msg else '?'}")


def check_p2p(rc, handle=None):
    if rc != TT_OK:
        msg = load().tt_p2p_last_error(handle)
        raise TTError(f"libttenv p2p error {rc}: {msg.decode() if msg else '?'}")


P2P_HANDLE_BYTES, P2P_MAX_RANKS = 128, 8


def default_params(variant=0):
    p = TTParams()
    check(load().tt_params_default(variant, C.byref(p)))
    return p
