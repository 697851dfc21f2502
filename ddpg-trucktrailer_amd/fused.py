"""Fused f32-accurate inference of the reference-shaped actor / critic (csrc/ttnet.hip, csrc/ttnet_split.hip) behind
torch tensors.

Used where no gradient is needed: the N-env `choose_action` of the rollout loop and the target-network
forward passes of `learn()`.  Networks of other shapes (or CPU tensors) report `supported(net) == False` and the
callers use the plain torch modules."""
import contextlib
import ctypes as C
import math

import torch

from ddpg_trucktrailer_amd import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def supported(net):
    if not all(hasattr(net, a) for a in ("fc1", "fc2", "bn1", "bn2")) or not (hasattr(net, "mu") or hasattr(net, "q")):
        return False          # a module of another structure: the callers use it through torch
    return (net.fc1.weight.is_cuda and net.fc1.weight.dtype == torch.float32 and tuple(net.fc1.weight.shape) == (400, 23)
            and tuple(net.fc2.weight.shape) == (300, 400) and net.bn1.eps == 1e-5 and net.bn2.eps == 1e-5)


def pack_and_sample(net, index, sample_args, cursor=None):
    """pack() and the replay draw described by `sample_args` (TrajectoryRing.sample_args) in ONE launch
    (tt_mlp_split_pack_and_sample): what opens a pipelined vector step.  cursor: TrajectoryRing.cursor() or None."""
    w = packed_weights_of(net, index)
    dev = net.fc2.weight.device
    L.check(L.load().tt_mlp_split_pack_and_sample(C.byref(w), 1 if hasattr(net, "action_value") else 0, C.c_void_p(w.split_ws),
                                                  C.byref(sample_args), C.byref(cursor) if cursor is not None else None,
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


def actor_act_ring(net, ring_view, weights, ou_state, act_scaled, seed, step=0, step_dev=None, theta=0.2, sigma=0.15, dt=1e-2,
                   high=math.pi / 4):
    """actor_act on the ring slot the device cursor names (tt_actor_act_ring): observations from slot t, stored action
    into slot t, noise restarted where slot t-1 says done."""
    dev = net.fc2.weight.device
    L.check(L.load().tt_actor_act_ring(int(ring_view.n_envs), C.byref(ring_view), C.byref(weights), _ptr(ou_state),
                                       int(seed) & (2 ** 64 - 1), int(step), _ptr(step_dev), float(theta * dt),
                                       float(sigma * math.sqrt(dt)), float(high), _ptr(act_scaled),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return act_scaled


def _fill_weights(net, w):
    head = net.mu if hasattr(net, "mu") else net.q
    for name, t in (("w1", net.fc1.weight), ("b1", net.fc1.bias), ("g1", net.bn1.weight), ("be1", net.bn1.bias),
                    ("w2", net.fc2.weight), ("b2", net.fc2.bias), ("g2", net.bn2.weight), ("be2", net.bn2.bias),
                    ("w3", head.weight), ("b3", head.bias)):
        assert t.is_contiguous()
        setattr(w, name, t.data_ptr())
    if hasattr(net, "action_value"):
        w.wa, w.ba = net.action_value.weight.data_ptr(), net.action_value.bias.data_ptr()
    w.in_dim, w.fc1_dims, w.fc2_dims = 23, 400, 300
    return w


def weights_of(net):
    """TTMlpWeights over the module's parameter storages (valid while the parameters are updated in place)."""
    cached = getattr(net, "_tt_weights", None)
    key = tuple(p.data_ptr() for p in net.parameters())
    if cached is not None and cached[0] == key:
        return cached[1]
    w = _fill_weights(net, L.TTMlpWeights())
    # workspace of the split-f16 kernel (csrc/ttnet_split.hip), re-packed from the weights by every call that uses it; one per
    # module, so two streams never share one (the learner's side stream runs other modules)
    net._tt_split_ws = torch.empty(int(L.load().tt_mlp_split_ws_bytes()), dtype=torch.uint8, device=net.fc2.weight.device)
    w.split_ws = net._tt_split_ws.data_ptr()
    net._tt_weights = (key, w)
    return w


def packed_weights_of(net, index, max_workgroups=None, capped_grids=None, two_images=None):
    """TTMlpWeights whose split-kernel image is kept current BY THE CALLER (pack()): one of two private workspaces per
    module (index 0 / 1), so a loop can fill the image for the next step while a forward still reads this step's.
    Forwards through it never re-pack and read nothing of the live parameters."""
    cache = net.__dict__.setdefault("_tt_packed", {})
    key = tuple(p.data_ptr() for p in net.parameters())
    hit = cache.get(index)
    if hit is None or hit[0] != key:
        # (parameter storage replaced: a new struct -- with everything the old one had been given, the second image
        # included, so that a pack() that follows writes the parity a later policy launch reads; a loop that captured
        # launches of the old struct re-captures: DDPGRollout watches the key through packed_key_of)
        old = hit[1] if hit is not None else None
        w = _fill_weights(net, L.TTMlpWeights())
        ws = torch.empty(int(L.load().tt_mlp_split_ws_bytes()), dtype=torch.uint8, device=net.fc2.weight.device)
        w.split_ws, w.ws_packed = ws.data_ptr(), 1
        hit = cache[index] = (key, w, ws)
        if old is not None:
            w.max_workgroups, w.capped_grids = old.max_workgroups, old.capped_grids
            two_images = two_images or bool(old.split_ws_alt)
    if max_workgroups is not None:
        hit[1].max_workgroups = int(max_workgroups)
    if capped_grids is not None:
        hit[1].capped_grids = int(capped_grids)
    if two_images and not hit[1].split_ws_alt:
        # a second image for odd steps (ring addressing: include/ttenv.h, tt_mlp_weights.split_ws_alt), kept with the struct
        alt = torch.empty(int(L.load().tt_mlp_split_ws_bytes()), dtype=torch.uint8, device=net.fc2.weight.device)
        hit[1].split_ws_alt = alt.data_ptr()
        cache[index] = hit + (alt,)
    return hit[1]


def packed_key_of(net):
    """What packed_weights_of() keys its structs on (the parameters' storages): a loop that captured launches of a struct
    compares this before it replays them."""
    return tuple(p.data_ptr() for p in net.parameters())


def pack(net, index, bump=None, cursor=None):
    """Write the split kernel's image of `net`'s CURRENT weights into its workspace `index` (tt_mlp_split_pack);
    bump: device int64 scalar incremented by the launch or None; cursor: TrajectoryRing.cursor() -- the launch then also
    writes the ring slots of the vector step it opens."""
    w = packed_weights_of(net, index)
    dev = net.fc2.weight.device
    L.check(L.load().tt_mlp_split_pack(C.byref(w), 1 if hasattr(net, "action_value") else 0, C.c_void_p(w.split_ws),
                                       _ptr(bump), C.byref(cursor) if cursor is not None else None,
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


@contextlib.contextmanager
def exact_f32(net):
    """Within the block, forwards of `net` run on the exact-f32 MFMA kernel whatever the batch (the split-f16 kernel's
    bit-reference; tests and A/B timing)."""
    w = weights_of(net)
    ws, w.split_ws = w.split_ws, None
    try:
        yield
    finally:
        w.split_ws = ws


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def actor_forward(net, obs, out=None):
    """ActorNetwork.forward without autograd: obs [n,23] f32 -> mu [n,1]."""
    n = obs.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=obs.device) if out is None else out
    L.check(L.load().tt_actor_forward(n, _ptr(obs), C.byref(weights_of(net)), _ptr(out), _stream(obs)))
    return out.view(n, 1)


def critic_forward(net, obs, action, out=None):
    """CriticNetwork.forward without autograd: obs [n,23], action [n,1] -> q [n,1]."""
    n = obs.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=obs.device) if out is None else out
    L.check(L.load().tt_critic_forward(n, _ptr(obs), _ptr(action), C.byref(weights_of(net)), _ptr(out), _stream(obs)))
    return out.view(n, 1)


def actor_act(net, obs, ou_state, act_raw, act_scaled, seed, step=0, step_dev=None, done_prev=None, mu_out=None,
              theta=0.2, sigma=0.15, dt=1e-2, high=math.pi / 4, weights=None):
    """choose_action + OU noise + clip*high for all rows in one launch (see include/ttenv.h: tt_actor_act).
    weights: a packed_weights_of() struct (the caller keeps its image current); default: re-pack on every call."""
    n = obs.shape[0]
    L.check(L.load().tt_actor_act(n, _ptr(obs), C.byref(weights if weights is not None else weights_of(net)), _ptr(ou_state), _ptr(done_prev),
                                  int(seed) & (2 ** 64 - 1), int(step), _ptr(step_dev), float(theta * dt),
                                  float(sigma * math.sqrt(dt)), float(high), _ptr(mu_out), _ptr(act_raw),
                                  _ptr(act_scaled), _stream(obs)))
    return act_scaled


def policy_kernel_info(n):
    """What the N-env policy forward executes on the matrix cores, for bench.py's roofline_mfma object."""
    waves = 4 * ((n + 127) // 128)
    return {"kernel": "k_mlp_split (choose_action for N envs; split-f16; the image's pack launch, ~4 us per step, not included)",
            # v_mfma_f32_32x32x16_f16, 32768 FLOP each: layer 2 = 25 k16 steps x 10 tiles x 3 products, layer 1 = 2 x 13 x 3
            "mfma_flop": waves * (750 + 78) * 32768.0,
            "note": "f16 MFMA FLOP executed (three f16 products per f32 product block, both layers) over the dense f16/bf16 "
                    "peak; algorithmic_* = the network's useful f32 FLOP"}
