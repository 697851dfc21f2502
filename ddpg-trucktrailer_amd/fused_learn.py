"""FusedLearner: Agent.learn() (DDPG/DDPG_agent.py:72-106) as five hand-written HIP launches after the sampling one
(csrc/ttlearn.hip) instead of ~140 autograd kernels, for the reference-shaped networks (23-400-300-1, LayerNorm) on a
GPU; with data-parallel ranks, eight launches and the two gradient all-reduces.

Same order of operations as the reference: TD target from the target nets -> critic MSE step -> actor step through
the ALREADY UPDATED critic -> soft update of both targets.  Same optimizer arithmetic (torch.optim.Adam with the
critic's weight decay folded into the gradient).  Parity with the torch path / the reference's fixture F5 is tested
in tests/test_gpu_fused_learn.py."""
import ctypes as C

import torch

from ddpg_trucktrailer_amd import _lib as L
from ddpg_trucktrailer_amd import fused

_ORDER = ("fc1.weight", "fc1.bias", "bn1.weight", "bn1.bias", "fc2.weight", "fc2.bias", "bn2.weight", "bn2.bias")
_FIELDS = ("w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "w3", "b3", "wa", "ba")


def _named(net):
    d = dict(net.named_parameters())
    head = "mu" if hasattr(net, "mu") else "q"
    names = list(_ORDER) + [head + ".weight", head + ".bias"]
    if hasattr(net, "action_value"):
        names += ["action_value.weight", "action_value.bias"]
    return [d[n] for n in names]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _no_sync():
    return None


class _DeviceMemory:
    """numel f32 of device memory owned by someone else, for torch.as_tensor (the CUDA array interface, version 2)."""

    def __init__(self, ptr, numel):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def _device_view(ptr, numel, dev, owner=None):
    if not ptr:
        raise RuntimeError("the peer-to-peer exchange returned no gradient buffer")
    t = torch.as_tensor(_DeviceMemory(ptr, numel), device=dev)
    assert t.data_ptr() == int(ptr) and t.numel() == numel and t.dtype == torch.float32, "torch copied the external buffer"
    t._tt_owner = owner                   # (the memory lives as long as the exchange handle does)
    return t


class _NetState:
    """Per-network device buffers: flat gradient (views per parameter, in tt_mlp_weights order), Adam moments."""

    def __init__(self, net, target, batch, dev):
        self.net, self.target = net, target
        self.params = _named(net)
        self.targets = _named(target) if target is not None else None
        self.critic = hasattr(net, "action_value")
        n = sum(p.numel() for p in self.params)
        f = dict(dtype=torch.float32, device=dev)
        self.m, self.v = torch.zeros(n, **f), torch.zeros(n, **f)
        self.ms, self.vs, off = [], [], 0
        for p in self.params:
            k = p.numel()
            self.ms.append(self.m[off:off + k]); self.vs.append(self.v[off:off + k])
            off += k
        self.gstruct = L.TTMlpWeights()
        self.gstruct.in_dim, self.gstruct.fc1_dims, self.gstruct.fc2_dims = 23, 400, 300
        self.bind_flat_grad(torch.zeros(n, **f))
        self.saved_t = dict(xh1=torch.empty((batch, 400), **f), h1=torch.empty((batch, 400), **f),
                            xh2=torch.empty((batch, 300), **f), h2=torch.empty((batch, 300), **f),
                            rstd1=torch.empty(batch, **f), rstd2=torch.empty(batch, **f))
        self.saved = L.TTMlpSaved(**{k: v.data_ptr() for k, v in self.saved_t.items()})
        cnt = len(self.params)
        arr = lambda ts: (C.c_void_p * cnt)(*[t.data_ptr() for t in ts])
        self.a_p, self.a_m, self.a_v = arr(self.params), arr(self.ms), arr(self.vs)
        self.a_t = arr(self.targets) if self.targets is not None else None
        self.a_n = (C.c_int32 * cnt)(*[p.numel() for p in self.params])
        self.count = cnt

    def bind_flat_grad(self, flat):
        """The flat gradient buffer (tt_mlp_weights order) the backward launches write and the optimizer launch / the gradient
        exchange read: a torch allocation, or a view of a site's buffer of the peer-to-peer exchange (FusedLearner.enable_p2p).
        Launches captured before a re-bind keep the old addresses: bind before capturing."""
        assert flat.numel() == sum(p.numel() for p in self.params) and flat.dtype == torch.float32
        self.flat_grad = flat
        self.grads, off = [], 0
        for p in self.params:
            k = p.numel()
            self.grads.append(flat[off:off + k].view_as(p))
            off += k
        for name, g in zip(_FIELDS, self.grads):
            setattr(self.gstruct, name, g.data_ptr())
        self.a_g = (C.c_void_p * len(self.params))(*[t.data_ptr() for t in self.grads])


class FusedLearner:
    def __init__(self, agent, batch_size, fc2_images=None):
        """fc2_images (None = on unless TT_LEARN_F32=1): the 400 x 300 products of learn() on the f16 MFMA from pre-split
        images of the four networks' fc2 (include/ttenv.h: tt_mlp_weights.fc2_img) that the optimizer launches keep current;
        off = every product on the exact-f32 MFMA straight from the weights."""
        import os
        assert fused.supported(agent.actor) and fused.supported(agent.critic)
        self.agent, self.B = agent, int(batch_size)
        dev = self.dev = agent.actor.fc1.weight.device
        self.lib = L.load()
        self.critic = _NetState(agent.critic, agent.target_critic, self.B, dev)
        self.actor = _NetState(agent.actor, agent.target_actor, self.B, dev)
        self.use_images = (os.environ.get("TT_LEARN_F32") != "1") if fc2_images is None else bool(fc2_images)
        # this learner's own weight structs of the four networks (the module-level ones of fused.weights_of serve inference)
        self._wstruct, self._img, self._img_seen = {}, {}, {}
        for net in (agent.actor, agent.critic, agent.target_actor, agent.target_critic):
            self._make_weights(net)
        for st in (self.actor, self.critic):
            st.images = L.TTFc2Images(net=self._img_ptr(st.net), target=self._img_ptr(st.target)) if self.use_images else None
        f = dict(dtype=torch.float32, device=dev)
        B = self.B
        self.ws_t = dict(dpre=torch.empty(B, **f), dz=torch.empty((B, 300), **f), dx2=torch.empty((B, 300), **f),
                         dy1=torch.empty((B, 400), **f), dx1=torch.empty((B, 400), **f))
        self.ws = L.TTMlpBwdWs(**{k: v.data_ptr() for k, v in self.ws_t.items()})
        # the actor's per-row gradients have a workspace of their own: its (unit) backward runs beside the critic's
        self.ws_actor_t = {k: torch.empty_like(v) for k, v in self.ws_t.items()}
        self.ws_actor = L.TTMlpBwdWs(**{k: v.data_ptr() for k, v in self.ws_actor_t.items()})
        self.mu_t, self.q_t, self.y, self.q, self.mu, self.q_pi, self.dq_da = (torch.empty(B, **f) for _ in range(7))
        self.step_dev = torch.zeros((), dtype=torch.int64, device=dev)      # learn() calls done (Adam's step count)
        # Adam's bias corrections of the running step, left by the launch that advances step_dev (tt_td_input.bias_corr_out)
        self.bias_corr = torch.zeros(8, dtype=torch.float32, device=dev)
        self.z_t = torch.empty((B, 300), **f)          # the target critic's state branch on s' (before the action enters)
        self.grad_sync_critic = self.grad_sync_actor = None
        self.p2p = None                    # the peer-to-peer gradient exchange (enable_p2p), a tt_p2p handle
        # learn()'s last two launches in ONE grid (tt_mlp_actor_tail): dQ/da is handed to the actor's weight-gradient workgroups in
        # device memory.  For the configurations learn() bounds (a small policy launch, several updates per step): its 200 waiting
        # workgroups would crowd a policy launch that owns 171 CUs.  Set by the loop (DDPGRollout), off by default.
        self.fuse_tail = False
        self.tail_words = torch.full((64 + 2 * 1024,), -1, dtype=torch.int32, device=dev)      # hints + one {step, dQ/da} word per row
        self.tail_gave_up_host = torch.zeros(2, dtype=torch.int32).pin_memory() if dev.type == "cuda" else None
        ga, gc = agent.actor.optimizer.param_groups[0], agent.critic.optimizer.param_groups[0]
        self.hyp_actor = (ga["lr"], ga["betas"][0], ga["betas"][1], ga["eps"], ga["weight_decay"])
        self.hyp_critic = (gc["lr"], gc["betas"][0], gc["betas"][1], gc["eps"], gc["weight_decay"])

    # ------------------------------------------------------------------------------------------------- fc2 images
    def _make_weights(self, net):
        w = fused._fill_weights(net, L.TTMlpWeights())
        if self.use_images:
            img = torch.zeros(int(self.lib.tt_mlp_fc2_image_bytes()), dtype=torch.uint8, device=self.dev)   # padding stays zero
            self._img[id(net)] = img
            w.fc2_img = img.data_ptr()
        self._wstruct[id(net)] = (tuple(p.data_ptr() for p in net.parameters()), w)
        self._img_seen[id(net)] = None

    def _img_ptr(self, net):
        t = self._img.get(id(net))
        return None if t is None else t.data_ptr()

    def w(self, net):
        """The learner's tt_mlp_weights of one of its four networks (with the fc2 image when images are on)."""
        key, w = self._wstruct[id(net)]
        if key != tuple(p.data_ptr() for p in net.parameters()):      # parameter storage replaced: rebuild
            self._make_weights(net)
            for st in (self.actor, self.critic):
                if self.use_images:
                    st.images = L.TTFc2Images(net=self._img_ptr(st.net), target=self._img_ptr(st.target))
            w = self._wstruct[id(net)][1]
        return w

    def images_current(self):
        """Have the four fc2 tensors been left alone by everything but this learner's own launches since their images were
        made?  (torch counts in-place writes per tensor; the learner's kernels update weights AND images together.)"""
        if not self.use_images:
            return True
        ag = self.agent
        return all(self._img_seen[id(n)] == (n.fc2.weight._version, n.fc2.weight.data_ptr())
                   for n in (ag.actor, ag.critic, ag.target_actor, ag.target_critic))

    def refresh_images(self, force=False):
        """(Re)make the image of every network whose fc2 was written by anyone else -- load_state_dict, a torch optimizer,
        the hard target copy -- since the last look.  Called at the top of every eager learn step; a loop that replays
        captured graphs calls it before it replays (DDPGRollout.run)."""
        if not self.use_images:
            return
        ag = self.agent
        for n in (ag.actor, ag.critic, ag.target_actor, ag.target_critic):
            seen = (n.fc2.weight._version, n.fc2.weight.data_ptr())
            if force or self._img_seen[id(n)] != seen:
                L.check(self.lib.tt_mlp_fc2_image_pack(C.byref(self.w(n)), self._stream()))
                self._img_seen[id(n)] = seen

    # -------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _fresh(self):
        if self.use_images and not torch.cuda.is_current_stream_capturing():
            self.refresh_images()          # (a capture holds launches only: its owner looks before it replays)

    def _fwd(self, net, obs, action, out, saved=None, dq_da=None):
        self._fresh()
        L.check(self.lib.tt_mlp_forward_save(self.B, 1 if action is not None else 0, _p(obs), _p(action),
                                             C.byref(self.w(net)), _p(out), C.byref(saved) if saved else None,
                                             _p(dq_da), self._stream()))

    def _bwd(self, st, mode, scale, obs, action, out, y=None, aux=None, td=None):
        self._fresh()
        L.check(self.lib.tt_mlp_backward(self.B, 1 if st.critic else 0, mode, float(scale), _p(obs), _p(action), None,
                                         _p(out), _p(y), _p(aux), C.byref(self.w(st.net)), C.byref(st.saved),
                                         C.byref(self.ws), C.byref(st.gstruct), C.byref(td) if td is not None else None,
                                         self._stream()))

    def _bwd_adam(self, st, hyp, tau, mode, scale, obs, action, out, y=None, aux=None, td=None):
        """_bwd + _adam in the backward's own two launches (include/ttenv.h: tt_mlp_backward_adam)."""
        lr, b1, b2, eps, wd = hyp
        L.check(self.lib.tt_mlp_backward_adam(self.B, 1 if st.critic else 0, mode, float(scale), _p(obs), _p(action), None,
                                              _p(out), _p(y), _p(aux), C.byref(self.w(st.net)),
                                              C.byref(st.saved), C.byref(self.ws), C.byref(st.gstruct), st.count, st.a_p,
                                              st.a_m, st.a_v, st.a_t, _p(self.step_dev), lr, b1, b2, eps, wd, tau,
                                              C.byref(td) if td is not None else None, self._stream()))
        if self.use_images:            # this (older) entry point does not maintain images: make them again before the next use
            self._img_seen[id(st.net)] = self._img_seen[id(st.target)] = None

    def _adam(self, st, hyp, tau):
        lr, b1, b2, eps, wd = hyp
        if self.p2p is not None:       # the mean of the ranks' gradients is formed INSIDE this launch (include/ttenv.h: tt_p2p_*)
            L.check_p2p(self.lib.tt_adam_soft_update_p2p(self.p2p, 0 if st.critic else 1, st.count, st.a_p, st.a_m, st.a_v, st.a_t, st.a_n,
                                                         _p(self.step_dev), lr, b1, b2, eps, wd, tau,
                                                         C.byref(st.images) if st.images is not None else None,
                                                         _p(self.bias_corr), self._stream()), self.p2p)
            return
        L.check(self.lib.tt_adam_soft_update(st.count, st.a_p, st.a_g, st.a_m, st.a_v, st.a_t, st.a_n, _p(self.step_dev),
                                             lr, b1, b2, eps, wd, tau, C.byref(st.images) if st.images is not None else None,
                                             _p(self.bias_corr), self._stream()))

    def enable_data_parallel(self, group=None):
        """Mean of the flat gradient buffers over the ranks at the reference's two optimizer sites (RCCL: one AVG
        all-reduce per site, straight on the flat buffer; other backends: SUM, then a divide)."""
        import torch.distributed as dist
        world = dist.get_world_size(group)
        avg = dist.get_backend(group) == "nccl"

        def sync(flat):
            if avg:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
                flat.div_(world)
        self.grad_sync_critic = lambda: sync(self.critic.flat_grad)
        self.grad_sync_actor = lambda: sync(self.actor.flat_grad)

    # ---- peer-to-peer gradient exchange ---------------------------------------------------------------------------
    def enable_p2p(self, group=None, timeout_s=None):
        """Data-parallel ranks WITHOUT collective launches on learn()'s chain (include/ttenv.h: tt_p2p_*): both flat gradient
        buffers move into a block of fine-grained device memory that the peers open through an IPC handle, and each rank's Adam
        launch reads every rank's gradients itself (sum in rank order / world: the same bits on every rank), behind a flag
        barrier in device memory.  The process group -- any backend -- only carries the handles (128 bytes per rank), once.  World size 1
        (no process group needed) runs the same launches against this rank's own block.  Call before anything is captured."""
        import os
        import torch.distributed as dist
        if timeout_s is None:      # (default here: 10 s -- the first launches of a process load its kernels, and the ranks do that at
            timeout_s = float(os.environ.get("TT_P2P_TIMEOUT_S", "10"))      # their own pace; the library's own default is 2 s)
        have = dist.is_available() and dist.is_initialized()
        world, rank = (dist.get_world_size(group), dist.get_rank(group)) if have else (1, 0)
        if world > L.P2P_MAX_RANKS:
            raise ValueError(f"the peer-to-peer exchange serves up to {L.P2P_MAX_RANKS} ranks (one node), not {world}")
        numel = (C.c_int32 * 2)(self.critic.flat_grad.numel(), self.actor.flat_grad.numel())
        h = C.c_void_p()
        index = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        L.check_p2p(self.lib.tt_p2p_create(index, rank, world, 2, numel, C.byref(h)))
        if timeout_s is not None:
            L.check_p2p(self.lib.tt_p2p_set_timeout(h, float(timeout_s)), h)
        mine = C.create_string_buffer(L.P2P_HANDLE_BYTES)
        L.check_p2p(self.lib.tt_p2p_export(h, mine), h)
        if world > 1:
            handles = [None] * world
            dist.all_gather_object(handles, bytes(mine.raw), group=group)
            for r, hb in enumerate(handles):
                if r != rank:
                    L.check_p2p(self.lib.tt_p2p_attach(h, r, C.create_string_buffer(hb, L.P2P_HANDLE_BYTES)), h)
            dist.barrier(group)            # every rank has opened every block before anyone launches into it
        self.p2p, self._p2p_group, self._p2p_world = h, group, world
        for site, st in enumerate((self.critic, self.actor)):
            st.bind_flat_grad(_device_view(self.lib.tt_p2p_grad(h, site), st.flat_grad.numel(), self.dev, owner=self))
        self.grad_sync_critic = self.grad_sync_actor = _no_sync       # (learn_batch's data-parallel order: separate Adam launches)

    def tail_gave_up(self):
        """0, or the learn step at which a weight-gradient workgroup of tt_mlp_actor_tail stopped waiting for dQ/da (that learn() is
        garbage).  Reads host memory only."""
        return int(self.tail_gave_up_host[0]) if self.tail_gave_up_host is not None else 0

    def p2p_gave_up(self):
        """0, or the learn step at which this rank's Adam launch stopped waiting for a peer's gradients (it then used whatever the
        buffers held: the ranks have diverged).  Reads host memory only."""
        return int(self.lib.tt_p2p_gave_up(self.p2p)) if self.p2p is not None else 0

    def p2p_reset(self):
        """After the learn-step counter was set back (a resume): arrival words of earlier runs must not satisfy new waits."""
        if self.p2p is None:
            return
        import torch.distributed as dist
        multi = self._p2p_world > 1
        torch.cuda.synchronize(self.dev)
        if multi:
            dist.barrier(self._p2p_group)
        L.check_p2p(self.lib.tt_p2p_reset(self.p2p, self._stream()), self.p2p)
        if multi:
            dist.barrier(self._p2p_group)

    # learn() in the three pieces the two gradient all-reduces cut it into (each piece is pure kernel launches on the
    # current stream, so a data-parallel loop can capture each as a hipGraph and keep only the collectives eager)
    def phase_a(self, states, actions, rewards, states_, done_u8, fuse_adam, window_dev=None, sample=None, image=None):
        """Forwards, TD target, critic backward (+ the critic's Adam/soft update in the same launch when fuse_adam).
        sample: a tt_sample_args (TrajectoryRing.sample_args) whose batch buffers ARE the five tensors given -- the forward
        launch then makes the replay draw itself (tt_mlp_forward_multi_sampled) instead of reading a batch drawn before.
        image: (packed actor weights struct, tt_ring_cursor whose k_dev is a snapshot word, that word's tensor) -- the critic's
        backward launch then also packs the vector step's policy image (tt_image_job; needs `sample`: the forward launch leaves
        the step number in the snapshot word)."""
        ag, B = self.agent, self.B
        self._fresh()
        # DDPG_agent.py:85-93 and :87, :101.  Only the target critic's LAST step needs the target actor's action (it enters
        # after LayerNorm2, networks.py:62-66), so four passes run side by side -- target actor on s', the target critic's
        # state branch on s', Q(s,a), mu(s), each filling 16 of the 256 CUs -- and the critic's backward then finishes
        # q'(s', mu'(s')) and the TD target.  The four are ONE launch (tt_mlp_forward_multi): a stream fork/join inside
        # a captured graph costs more than the kernel it would hide
        def ptr(t):
            return None if t is None else t.data_ptr()
        jobs = (L.TTFwdJob * 4)()
        for j, (net, crit, obs, act, out, saved, zst) in enumerate((
                (ag.target_actor, 0, states_, None, self.mu_t, None, None),
                (ag.target_critic, 1, states_, None, None, None, self.z_t),
                (ag.critic, 1, states, actions, self.q, self.critic.saved, None),
                (ag.actor, 0, states, None, self.mu, self.actor.saved, None))):
            jobs[j].critic, jobs[j].obs, jobs[j].action = crit, ptr(obs), ptr(act)
            jobs[j].w, jobs[j].out = C.pointer(self.w(net)), ptr(out)
            jobs[j].saved = C.pointer(saved) if saved is not None else None
            jobs[j].dq_da, jobs[j].z_state = None, ptr(zst)
        if sample is not None:
            assert (sample.s_out, sample.a_out, sample.s2_out) == (states.data_ptr(), actions.data_ptr(), states_.data_ptr())
            L.check(self.lib.tt_mlp_forward_multi_sampled(B, 4, jobs, C.byref(sample), _p(image[2]) if image is not None else None,
                                                          self._stream()))
        else:
            L.check(self.lib.tt_mlp_forward_multi(B, 4, jobs, self._stream()))
        # critic step (DDPG_agent.py:95-98); its backward launch first finishes q'(s', mu'(s')) and the TD target for its
        # rows (tt_td_input: what tt_critic_head_td does as a launch of its own)
        td = L.TTTdInput(z_state=self.z_t.data_ptr(), mu_target=self.mu_t.data_ptr(),
                         target_critic=C.pointer(self.w(ag.target_critic)), reward=rewards.data_ptr(),
                         done=done_u8.data_ptr(), gamma=float(ag.gamma), y_out=self.y.data_ptr(),
                         q_out=self.q_t.data_ptr(), step_dev=self.step_dev.data_ptr(),
                         window_dev=window_dev.data_ptr() if window_dev is not None else None,
                         bias_corr_out=self.bias_corr.data_ptr(), adam_beta1=self.hyp_critic[1], adam_beta2=self.hyp_critic[2])
        # ... and, on other workgroups of the same launch, the ACTOR's per-row backward for a unit gradient: it is linear in
        # the row's d(loss)/d(pre-tanh), which needs the updated critic and is applied in phase_b (include/ttenv.h)
        L.check(self.lib.tt_mlp_backward_rows_pair(B, 2.0 / B, _p(self.q), C.byref(self.w(ag.critic)),
                                                   C.byref(self.critic.saved), C.byref(self.ws), C.byref(td), _p(self.mu),
                                                   C.byref(self.w(ag.actor)), C.byref(self.actor.saved),
                                                   C.byref(self.ws_actor),
                                                   C.byref(L.TTImageJob(C.pointer(image[0]), C.pointer(image[1]))) if image is not None else None,
                                                   self._stream()))
        self._weights(self.critic, self.hyp_critic, ag.tau, states, actions, self.ws, adam=fuse_adam)

    def _weights(self, st, hyp, tau, obs, action, ws, adam, row=None):
        """The weight-gradient launch (tt_mlp_backward_weights), with Adam + soft update in it when `adam`."""
        lr, b1, b2, eps, wd = hyp
        dq, mu, sc = row if row is not None else (None, None, 1.0)
        L.check(self.lib.tt_mlp_backward_weights(self.B, 1 if st.critic else 0, _p(obs), _p(action), C.byref(st.saved), C.byref(ws),
                                                 C.byref(st.gstruct), _p(dq), _p(mu), float(sc), st.count if adam else 0,
                                                 st.a_p, st.a_m, st.a_v, st.a_t, _p(self.step_dev), lr, b1, b2, eps, wd, tau,
                                                 C.byref(st.images) if (adam and st.images is not None) else None,
                                                 _p(self.bias_corr), self._stream()))

    def phase_b(self, states, separate_adam):
        """[critic Adam/soft update when not already applied,] then the actor step through the UPDATED critic
        (DDPG_agent.py:100-104): Q(s, mu(s)) with dQ/da (a critic forward), and the actor's weight gradients from its unit
        per-row backward scaled by -(1/B) dQ/da (1 - mu^2) per row (+ its Adam unless separate_adam)."""
        ag, B = self.agent, self.B
        if separate_adam:
            self._adam(self.critic, self.hyp_critic, ag.tau)
        elif self.fuse_tail:
            self._fresh()
            st = self.actor
            lr, b1, b2, eps, wd = self.hyp_actor
            L.check(self.lib.tt_mlp_actor_tail(B, _p(states), _p(self.mu), C.byref(self.w(ag.critic)), _p(self.q_pi), _p(self.dq_da),
                                               C.byref(st.saved), C.byref(self.ws_actor), C.byref(st.gstruct), -1.0 / B, st.count,
                                               st.a_p, st.a_m, st.a_v, st.a_t, _p(self.step_dev), lr, b1, b2, eps, wd, ag.tau,
                                               C.byref(st.images) if st.images is not None else None, _p(self.bias_corr),
                                               _p(self.tail_words), C.c_void_p(self.tail_gave_up_host.data_ptr()), self._stream()))
            return
        self._fwd(ag.critic, states, self.mu, self.q_pi, dq_da=self.dq_da)
        self._weights(self.actor, self.hyp_actor, ag.tau, states, None, self.ws_actor, adam=not separate_adam,
                      row=(self.dq_da, self.mu, -1.0 / B))

    def phase_c(self):
        self._adam(self.actor, self.hyp_actor, self.agent.tau)

    def learn_batch(self, states, actions, rewards, states_, done_u8, window_dev=None, sample=None, image=None):
        """states, states_ [B,23] f32; actions [B,1] f32; rewards [B] f32; done_u8 [B] uint8 -- all contiguous.
        window_dev: device int64 advanced by the critic's backward launch (a pipelined loop's sampling window).
        sample: see phase_a (the five tensors are then the draw's batch buffers, filled by learn()'s first launch)."""
        assert states.shape[0] == self.B and done_u8.dtype == torch.uint8
        dp = self.grad_sync_critic is not None
        assert image is None or sample is not None
        self.phase_a(states, actions, rewards, states_, done_u8, fuse_adam=not dp, window_dev=window_dev, sample=sample, image=image)
        if dp:
            self.grad_sync_critic()
        self.phase_b(states, separate_adam=dp)
        if dp:
            self.grad_sync_actor()
            self.phase_c()

    # ---- checkpoint ---------------------------------------------------------------------------------------------
    def state_dict(self):
        """Adam moments of both networks (flat, tt_mlp_weights order) and the learn-step count."""
        return {"step": int(self.step_dev.item()),
                "actor": {"m": self.actor.m.cpu(), "v": self.actor.v.cpu()},
                "critic": {"m": self.critic.m.cpu(), "v": self.critic.v.cpu()}}

    def load_state_dict(self, sd):
        for name in ("actor", "critic"):
            st = getattr(self, name)
            st.m.copy_(sd[name]["m"]); st.v.copy_(sd[name]["v"])
        self.step_dev.fill_(int(sd["step"]))
        self.tail_words.fill_(-1)          # (tt_mlp_actor_tail: words of an earlier run must not match a step number set back)
        self.p2p_reset()

    # ---- checkpoint interoperability with the torch optimizers ------------------------------------------
    def export_to_optimizers(self):
        step = self.step_dev.to(torch.float32).clone()
        for st, opt in ((self.actor, self.agent.actor.optimizer), (self.critic, self.agent.critic.optimizer)):
            for p, m, v in zip(st.params, st.ms, st.vs):
                opt.state[p] = {"step": step.clone(), "exp_avg": m.view_as(p).clone(), "exp_avg_sq": v.view_as(p).clone()}

    def import_from_optimizers(self):
        for st, opt in ((self.actor, self.agent.actor.optimizer), (self.critic, self.agent.critic.optimizer)):
            for p, m, v in zip(st.params, st.ms, st.vs):
                s = opt.state.get(p)
                if s:
                    m.copy_(s["exp_avg"].reshape(-1)); v.copy_(s["exp_avg_sq"].reshape(-1))
                    self.step_dev.fill_(int(float(s["step"])))
