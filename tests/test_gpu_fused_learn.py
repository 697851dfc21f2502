"""GPU (-m gpu): the hand-fused learn() kernels (csrc/ttlearn.hip) against torch autograd on the same nets and
against fixture F5 (the reference's own learn()).  Tolerances as in tests/test_learner.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(dev, seed=0, B=256):
    import torch
    from test_gpu_fused_net import _nets
    actor, critic = _nets(dev, seed)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    s = torch.rand((B, 23), device=dev, generator=g) * 2 - 1
    a = torch.rand((B, 1), device=dev, generator=g) * 2.4 - 1.2
    return actor, critic, s, a, g


@pytest.mark.parametrize("B", [256, 100, 16])
def test_forward_save_and_backward_match_autograd(gpu_device, B):
    import ctypes as C
    import torch
    from ddpg_trucktrailer_amd import _lib as L, fused
    from ddpg_trucktrailer_amd.fused_learn import _NetState, _p
    actor, critic, s, a, g = _setup(gpu_device, seed=B, B=B)
    lib = L.load()
    f = dict(dtype=torch.float32, device=gpu_device)
    ws_t = dict(dpre=torch.empty(B, **f), dz=torch.empty((B, 300), **f), dx2=torch.empty((B, 300), **f),
                dy1=torch.empty((B, 400), **f), dx1=torch.empty((B, 400), **f))
    ws = L.TTMlpBwdWs(**{k: v.data_ptr() for k, v in ws_t.items()})
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for net, act in ((actor, None), (critic, a)):
        st = _NetState(net, None, B, gpu_device)
        out = torch.empty(B, **f)
        dq_da = torch.empty(B, **f) if act is not None else None
        L.check(lib.tt_mlp_forward_save(B, 1 if act is not None else 0, _p(s), _p(act), C.byref(fused.weights_of(net)), _p(out),
                                        C.byref(st.saved), _p(dq_da), stream))
        a_req = act.clone().requires_grad_(True) if act is not None else None
        ref = net(s) if act is None else net(s, a_req)
        assert (out.view(-1, 1) - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
        # saved activations = what autograd would keep
        x1 = net.fc1(s); h1 = torch.relu(net.bn1(x1))
        xh1 = (x1 - x1.mean(1, keepdim=True)) * torch.rsqrt(x1.var(1, unbiased=False, keepdim=True) + 1e-5)
        assert (st.saved_t["xh1"] - xh1).abs().max().item() <= 2e-5 and (st.saved_t["h1"] - h1).abs().max().item() <= 2e-5
        d_out = torch.randn(B, generator=g, **f)
        net.zero_grad(set_to_none=True)
        ref.backward(d_out.view(-1, 1))
        if act is not None:
            assert (dq_da - torch.autograd.grad(net(s, a_req).sum(), a_req)[0].view(-1)).abs().max().item() <= 2e-5
        L.check(lib.tt_mlp_backward(B, 1 if act is not None else 0, 0, 1.0, _p(s), _p(act), _p(d_out), _p(out), None, None,
                                    C.byref(fused.weights_of(net)), C.byref(st.saved), C.byref(ws), C.byref(st.gstruct), None, None, stream))
        for p, gk in zip(st.params, st.grads):
            scale = max(1e-3, p.grad.abs().max().item())
            assert (gk - p.grad).abs().max().item() <= 3e-5 * scale + 1e-6, (tuple(p.shape), (gk - p.grad).abs().max().item(), scale)


def _agent(dev, z, capturable=False):
    from ddpg_trucktrailer_amd.agent import Agent
    from test_learner import _load_init
    a = Agent(alpha=1e-4, beta=1e-3, input_dims=(23,), tau=1e-3, n_actions=1, batch_size=256, device=dev, replay=False,
              capturable=capturable)
    _load_init(a, z)
    return a


def _ln_params(fl):
    """The LayerNorm / action_value parameters the NEXT forward will use (cloned: the optimizer launch updates them in place)."""
    out = []
    for st in (fl.critic, fl.actor):
        net = st.net
        p = [net.bn1.weight, net.bn1.bias, net.bn2.weight, net.bn2.bias]
        if st.critic:
            p += [net.action_value.weight, net.action_value.bias]
        out.append([t.detach().clone() for t in p])
    return out


def _relu_margin(fl, actions, params=None):
    """Smallest |pre-activation| in front of any ReLU of the critic and the actor over the batch, from what the last forward
    saved (x-hat * gamma + beta [+ action_value(a)]); params: _ln_params() taken BEFORE that forward's learn() call when the
    optimizer has already run (default: the live parameters -- right between phase_a and the optimizer launches).  A unit closer to zero than the forward's own rounding (~1e-6, measured
    between two builds of the same kernels) can fall on the other side than in the reference: its row's gradient then changes
    by that unit's whole contribution -- ~6e-5 of a tensor's scale where all other errors are ~1e-6 (tools/dbg/grad_margin.py).
    The parity tests keep their tight bound unless such a unit is PRESENT in the very forward they check."""
    params = _ln_params(fl) if params is None else params
    worst = float("inf")
    for st, p in zip((fl.critic, fl.actor), params):
        z1 = st.saved_t["xh1"] * p[0] + p[1]
        z2 = st.saved_t["xh2"] * p[2] + p[3]
        if st.critic:
            z2 = z2 + actions.view(-1, 1) * p[4].view(1, -1) + p[5]
        worst = min(worst, z1.abs().min().item(), z2.abs().min().item())
    return worst


_RELU_BOUNDARY = 3e-6


@pytest.mark.parametrize("images", [True, False])
def test_fused_learn_matches_reference_fixture_and_torch_path(gpu_device, images):
    """images: learn()'s 400x300 products on the f16 MFMA from pre-split fc2 images (the default) / on the exact-f32 MFMA."""
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from test_learner import _batch, _check_snapshot
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    fused_agent, torch_agent = _agent(gpu_device, z), _agent(gpu_device, z)
    s, a, r, s2, d = _batch(z, gpu_device)
    fl = FusedLearner(fused_agent, 256, fc2_images=images)
    assert fl.use_images == images
    d8 = d.to(torch.uint8)
    fl.learn_batch(s, a, r, s2, d8)
    torch_agent.learn_batch(s, a, r, s2, d)
    _check_snapshot(fused_agent, z, "after1", 1e-5)                 # vs the REFERENCE's learn()
    assert (fl.y - torch.tensor(z["target_y"], device=gpu_device)).abs().max().item() <= 1e-5 * np.abs(z["target_y"]).max()
    boundary = False
    for _ in range(2):
        before = _ln_params(fl)
        fl.learn_batch(s, a, r, s2, d8)
        torch_agent.learn_batch(s, a, r, s2, d)
        boundary |= _relu_margin(fl, a, before) < _RELU_BOUNDARY
    _check_snapshot(fused_agent, z, "after3", 4e-5)
    for name in ("actor", "critic", "target_actor", "target_critic"):
        for (k, x), y in zip(getattr(fused_agent, name).state_dict().items(), getattr(torch_agent, name).state_dict().values()):
            assert (x - y).abs().max().item() <= 4e-5 * max(1e-1, y.abs().max().item()) + 1e-6, (name, k)
    assert int(fl.step_dev.item()) == 3
    # Adam state round trip with the torch optimizers (checkpoint interoperability)
    fl.export_to_optimizers()
    st = fused_agent.critic.optimizer.state[fused_agent.critic.fc2.weight]
    ref = torch_agent.critic.optimizer.state[torch_agent.critic.fc2.weight]
    assert float(st["step"]) == 3
    if boundary:    # a ReLU unit within rounding of zero in one of these forwards (see _relu_margin): scale-relative bound
        assert (st["exp_avg"] - ref["exp_avg"]).abs().max().item() <= 1e-3 * ref["exp_avg"].abs().max().item()
    else:
        assert torch.allclose(st["exp_avg"], ref["exp_avg"], rtol=1e-3, atol=1e-7)
    fl2 = FusedLearner(fused_agent, 256)
    fl2.import_from_optimizers()
    assert int(fl2.step_dev.item()) == 3 and torch.equal(fl2.critic.m, fl.critic.m)


@pytest.mark.parametrize("images", [True, False])
def test_fused_gradients_and_losses_match_the_reference(gpu_device, images):
    """FusedLearner's flat gradient buffers (what RCCL all-reduces) at both optimizer sites of learn() steps 1..3 against
    the .grad the reference's own learn() held at its optimizer.step() calls (fixture F5 grad<i>/..., DDPG_agent.py:95-104),
    3e-5 relative to each tensor's largest gradient; critic loss from the fused q / y, actor loss from Q(s, mu(s))."""
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner, _ORDER
    from test_learner import _batch, _check_grads
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    agent = _agent(gpu_device, z)
    s, a, r, s2, d = _batch(z, gpu_device)
    d8 = d.to(torch.uint8)
    fl = FusedLearner(agent, 256, fc2_images=images)

    def named(st):
        head = "q" if st.critic else "mu"
        names = list(_ORDER) + [head + ".weight", head + ".bias"] + (["action_value.weight", "action_value.bias"] if st.critic else [])
        return [(n, g.clone()) for n, g in zip(names, st.grads)]
    for i in (1, 2, 3):
        # the data-parallel pieces leave each site's gradient in place before its optimizer launch (phase_a / phase_b)
        fl.phase_a(s, a, r, s2, d8, fuse_adam=False)
        # 3e-5 of each tensor's scale (measured: <= 1e-6) -- 1e-4 only where this very forward holds a ReLU unit within
        # rounding of zero (_relu_margin; with the f16-image products that happens at step 3: one unit of row 112)
        rtol = 3e-5 if _relu_margin(fl, a) >= _RELU_BOUNDARY else 1e-4
        _check_grads(z, i, "critic", named(fl.critic), rtol)
        loss_c = torch.mean((fl.q - fl.y) ** 2).item()
        assert abs(loss_c - float(z[f"loss{i}/critic"])) <= 1e-5 * float(z[f"loss{i}/critic"])
        fl.phase_b(s, separate_adam=True)
        _check_grads(z, i, "actor", named(fl.actor), rtol)
        loss_a = -fl.q_pi.mean().item()
        assert abs(loss_a - float(z[f"loss{i}/actor"])) <= 2e-5 * max(0.1, abs(float(z[f"loss{i}/actor"])))
        fl.phase_c()
    from test_learner import _check_snapshot
    _check_snapshot(agent, z, "after3", 4e-5)


def test_fused_learn_in_hipgraph(gpu_device):
    """Captured once, replayed: every replay advances the step counter and keeps matching the eager fused path."""
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from test_learner import _batch
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    eager_agent, graph_agent = _agent(gpu_device, z), _agent(gpu_device, z)
    s, a, r, s2, d = _batch(z, gpu_device)
    d8 = d.to(torch.uint8)
    fe, fg = FusedLearner(eager_agent, 256), FusedLearner(graph_agent, 256)
    fe.learn_batch(s, a, r, s2, d8); fg.learn_batch(s, a, r, s2, d8)          # warm-up (also sets kernel attributes)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fg.learn_batch(s, a, r, s2, d8)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay(); fe.learn_batch(s, a, r, s2, d8)
    torch.cuda.synchronize()
    assert int(fg.step_dev.item()) == int(fe.step_dev.item()) == 4
    for name in ("actor", "critic", "target_actor", "target_critic"):
        for x, y in zip(getattr(eager_agent, name).state_dict().values(), getattr(graph_agent, name).state_dict().values()):
            assert torch.equal(x, y), name


def test_adam_inside_weight_gradient_launch_is_bit_identical(gpu_device):
    """Single-rank learn() applies Adam + soft update inside k_bwd_weights (tt_mlp_backward_adam); ranks that all-reduce
    their gradients run tt_mlp_backward, the collective, tt_adam_soft_update.  Same arithmetic: identical bits."""
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from test_learner import _batch
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    one, two = _agent(gpu_device, z), _agent(gpu_device, z)
    s, a, r, s2, d = _batch(z, gpu_device)
    d8 = d.to(torch.uint8)
    f1, f2 = FusedLearner(one, 256), FusedLearner(two, 256)
    f2.grad_sync_critic = f2.grad_sync_actor = lambda: None          # a no-op "all-reduce": forces the separate launches
    for _ in range(3):
        f1.learn_batch(s, a, r, s2, d8); f2.learn_batch(s, a, r, s2, d8)
    torch.cuda.synchronize()
    for name in ("actor", "critic", "target_actor", "target_critic"):
        for x, y in zip(getattr(one, name).state_dict().values(), getattr(two, name).state_dict().values()):
            assert torch.equal(x, y), name
    assert torch.equal(f1.critic.m, f2.critic.m) and torch.equal(f1.actor.v, f2.actor.v)
    assert torch.equal(f1.critic.flat_grad, f2.critic.flat_grad)     # the gradients are still written


def test_target_critic_in_two_pieces(gpu_device):
    """tt_critic_state_forward + tt_critic_head_td (the state branch next to the target actor, then q' and the TD target
    in one small launch) against the one-piece critic forward + tt_td_target."""
    import ctypes as C
    import torch
    from ddpg_trucktrailer_amd import _lib as L, fused
    from ddpg_trucktrailer_amd.fused_learn import _p
    B = 200
    actor, critic, s, a, g = _setup(gpu_device, seed=9, B=B)
    lib = L.load()
    f = dict(dtype=torch.float32, device=gpu_device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    r = torch.randn(B, generator=g, **f)
    done = (torch.rand(B, device=gpu_device, generator=g) < 0.3).to(torch.uint8)
    z, y, q, y_ref, q_ref = torch.empty((B, 300), **f), torch.empty(B, **f), torch.empty(B, **f), torch.empty(B, **f), torch.empty(B, **f)
    w = fused.weights_of(critic)
    L.check(lib.tt_critic_state_forward(B, _p(s), C.byref(w), _p(z), stream))
    step = torch.zeros((), dtype=torch.int64, device=gpu_device)
    L.check(lib.tt_critic_head_td(B, _p(z), _p(a), C.byref(w), _p(r), _p(done), 0.99, _p(y), _p(q), _p(step), stream))
    L.check(lib.tt_mlp_forward_save(B, 1, _p(s), _p(a), C.byref(w), _p(q_ref), None, None, stream))
    L.check(lib.tt_td_target(B, _p(r), _p(q_ref), _p(done), 0.99, _p(y_ref), None, stream))
    with torch.no_grad():
        z_ref = critic.bn2(critic.fc2(torch.relu(critic.bn1(critic.fc1(s)))))
        q_t = critic(s, a).view(-1)
    assert (z - z_ref).abs().max().item() <= 2e-5
    assert (q - q_ref).abs().max().item() <= 1e-5 and (q - q_t).abs().max().item() <= 2e-5 * max(1.0, q_t.abs().max().item())
    assert (y - y_ref).abs().max().item() <= 1e-5 and int(step.item()) == 1
    assert torch.equal(y[done.bool()], r[done.bool()])               # critic_value_[done] = 0 (DDPG_agent.py:89)


def test_fc2_images_follow_the_weights(gpu_device):
    """The pre-split f16 images of fc2 (tt_mlp_weights.fc2_img) that learn()'s kernels read instead of w2: kept current by the
    optimizer launches element by element (Adam inside the weight-gradient launch, and as its own launch after an
    all-reduce), so after any number of steps they equal an image made from scratch, bit for bit; pieces h + m reproduce
    64 w to 2^-22; a write to fc2 by anyone else (torch) is noticed and the image made again."""
    import ctypes as C
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd import _lib as L
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from test_learner import _batch
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    s, a, r, s2, d = _batch(z, gpu_device)
    d8 = d.to(torch.uint8)
    lib = L.load()
    nbytes = int(lib.tt_mlp_fc2_image_bytes())
    FWD, BWD = 20 * 13 * 512, 28 * 10 * 512            # halves per plane: MFMA fragments of 64 lanes x 8 halves (ttlearn.hip)
    assert nbytes == 2 * 2 * (FWD + BWD)
    nn, kk = np.meshgrid(np.arange(300), np.arange(400), indexing="ij")
    fwd_idx = torch.tensor((((nn >> 4) * 13 + (kk >> 5)) * 64 + ((kk >> 3) & 3) * 16 + (nn & 15)) * 8 + (kk & 7), device=gpu_device)
    tile = (kk >> 6) * 4 + (kk & 3)
    bwd_idx = torch.tensor(((tile * 10 + (nn >> 5)) * 64 + ((nn >> 3) & 3) * 16 + ((kk >> 2) & 15)) * 8 + (nn & 7), device=gpu_device)

    def scratch(fl, net):
        w = L.TTMlpWeights()
        C.memmove(C.byref(w), C.byref(fl.w(net)), C.sizeof(w))
        img = torch.zeros(nbytes, dtype=torch.uint8, device=gpu_device)
        w.fc2_img = img.data_ptr()
        L.check(lib.tt_mlp_fc2_image_pack(C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return img

    for dp_style in (False, True):
        agent = _agent(gpu_device, z)
        fl = FusedLearner(agent, 256, fc2_images=True)
        if dp_style:                                   # separate Adam launches, as after a gradient all-reduce
            fl.grad_sync_critic = fl.grad_sync_actor = lambda: None
        for _ in range(4):
            fl.learn_batch(s, a, r, s2, d8)
        torch.cuda.synchronize()
        assert fl.images_current()
        for net in (agent.actor, agent.critic, agent.target_actor, agent.target_critic):
            kept, fresh = fl._img[id(net)], scratch(fl, net)
            halves = kept.view(torch.float16)
            fwd = FWD
            is_target = net in (agent.target_actor, agent.target_critic)
            if is_target:                              # a target's image serves forwards only: its [k][n] half is not maintained
                assert torch.equal(halves[:2 * fwd], fresh.view(torch.float16)[:2 * fwd]), "target image differs"
            else:
                assert torch.equal(kept, fresh), "maintained image differs from one made from scratch"
            w2 = net.fc2.weight.detach()
            hp, mp = halves[:FWD], halves[FWD:2 * FWD]
            h, m = hp[fwd_idx].float(), mp[fwd_idx].float()
            assert ((h + m) / 64 - w2).abs().max().item() <= 2 ** -22 * w2.abs().max().item()
            rest = hp.clone(); rest[fwd_idx.reshape(-1)] = 0
            assert not rest.any(), "padding of the forward plane is not zero"
            if not is_target:                          # the backward orientation holds the same numbers
                th = halves[2 * FWD:2 * FWD + BWD]
                assert torch.equal(th[bwd_idx], hp[fwd_idx])
                rest = th.clone(); rest[bwd_idx.reshape(-1)] = 0
                assert not rest.any()
        # someone else writes fc2: noticed, image made again before the next step
        with torch.no_grad():
            agent.critic.fc2.weight.mul_(1.01)
        assert not fl.images_current()
        fl.learn_batch(s, a, r, s2, d8)
        assert fl.images_current()
        assert torch.equal(fl._img[id(agent.critic)], scratch(fl, agent.critic))


@pytest.mark.parametrize("side", [0, 300])
def test_learn_with_the_draw_made_by_its_first_launch(gpu_device, side):
    """tt_mlp_forward_multi_sampled: learn()'s first launch makes the replay draw itself (ring rows read in place, the five batch
    buffers filled on the way) == tt_ring_sample followed by the same learn(), bit for bit -- batch buffers, weights, Adam
    state; with and without side (expert) tuples in the draw; the window arguments of the pipelined loop included."""
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from ddpg_trucktrailer_amd.replay_buffer import TrajectoryRing
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    dev = gpu_device
    g = torch.Generator(device=dev); g.manual_seed(3)
    n, slots, B = 512, 16, 256
    ring = TrajectoryRing(n, slots, 23, dev)
    ring.obs.copy_(torch.rand(ring.obs.shape, device=dev, generator=g) * 2 - 1)
    ring.act.copy_(torch.rand(ring.act.shape, device=dev, generator=g) * 2 - 1)
    ring.rew.copy_(torch.rand(ring.rew.shape, device=dev, generator=g) * 10 - 5)
    ring.done.copy_((torch.rand(ring.done.shape, device=dev, generator=g) < 0.05).to(torch.uint8))
    ring.k = 37; ring.k_dev.fill_(37)
    if side:
        f = lambda *s: torch.rand(s, device=dev, generator=g)
        ring.load_side(f(side, 23), f(side, 1), f(side), f(side, 23), (f(side) < 0.1))
    agents = [_agent(dev, z), _agent(dev, z)]
    learners = [FusedLearner(a, B) for a in agents]
    for step in range(3):
        kw = dict(seed=1234 + step) if step < 2 else dict(seed=99, reserve=2, lag=1)
        s, a, r, s2, d = ring.sample_fused(B, done_as_bool=False, **kw)
        drawn = [t.clone() for t in (s, a, r, s2, d)]
        learners[0].learn_batch(s, a, r, s2, d)
        for t in ring._batch_bufs(B)[:5]:
            t.zero_()                                    # the sampled launch must fill them itself
        args = ring.sample_args(B, **kw)
        s, a, r, s2, d = ring._batch_bufs(B)[:5]
        learners[1].learn_batch(s, a, r, s2, d, sample=args)
        torch.cuda.synchronize()
        for x, y in zip(drawn, (s, a, r, s2, d)):
            assert torch.equal(x, y), step
        for name in ("actor", "critic", "target_actor", "target_critic"):
            for x, y in zip(getattr(agents[0], name).state_dict().values(), getattr(agents[1], name).state_dict().values()):
                assert torch.equal(x, y), (step, name)
        assert torch.equal(learners[0].critic.m, learners[1].critic.m) and torch.equal(learners[0].actor.v, learners[1].actor.v)


def test_policy_image_packed_by_learns_second_launch(gpu_device):
    """tt_image_job: the critic-backward launch of a learn() carries the pack of the vector step's policy image on workgroups of
    its own.  The image (of the step's parity), the ring cursor and the image epoch it leaves == what tt_mlp_split_pack makes
    of the actor's weights as they were BEFORE this learn() (the actor is written by learn()'s last launch), bit for bit; the
    step number comes from the snapshot the forward launch took, not from the counter this launch advances."""
    import ctypes as C
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd import _lib as L, fused
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from ddpg_trucktrailer_amd.replay_buffer import TrajectoryRing
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    dev = gpu_device
    g = torch.Generator(device=dev); g.manual_seed(5)
    n, slots, B = 512, 16, 256
    ring = TrajectoryRing(n, slots, 23, dev)
    ring.obs.copy_(torch.rand(ring.obs.shape, device=dev, generator=g) * 2 - 1)
    ring.act.copy_(torch.rand(ring.act.shape, device=dev, generator=g) * 2 - 1)
    ring.rew.copy_(torch.rand(ring.rew.shape, device=dev, generator=g))
    agent = _agent(dev, z)
    fl = FusedLearner(agent, B)
    snap = torch.zeros((), dtype=torch.int64, device=dev)
    window = torch.zeros((), dtype=torch.int64, device=dev)
    w = fused.packed_weights_of(agent.actor, 0, 192, 4, two_images=True)
    images = agent.actor._tt_packed[0][2:4]                      # (even, odd) image buffers of the struct
    cur = L.TTRingCursor(snap.data_ptr(), slots, 0, ring.cursor_dev.data_ptr())
    for k in (37, 38):                                           # an odd and an even step
        ring.k = k; ring.k_dev.fill_(k); window.fill_(k)
        before = _agent(dev, z).actor                          # a second module with the actor's weights of this moment
        before.load_state_dict(agent.actor.state_dict())
        args = ring.sample_args(B, seed=7 + k, k_dev=window, reserve=2, lag=1)
        s, a, r, s2, d = ring._batch_bufs(B)[:5]
        for t in images:
            t.zero_()
        fl.learn_batch(s, a, r, s2, d, window_dev=window, sample=args, image=(w, cur, snap))
        torch.cuda.synchronize()
        assert int(snap.item()) == k and int(window.item()) == k + 1          # the launch read k, then moved the window on
        ref = torch.empty_like(images[0])
        L.check(L.load().tt_mlp_split_pack(C.byref(fused.packed_weights_of(before, 0)), 0, C.c_void_p(ref.data_ptr()), None, None,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert torch.equal(images[k & 1], ref), k
        assert not torch.equal(fused.actor_forward(agent.actor, s).view(-1), fused.actor_forward(before, s).view(-1))   # learn() moved it
        cw = ring.cursor_dev.cpu().tolist()
        assert cw[4 + 4 * (k & 1): 8 + 4 * (k & 1)] == [k % slots, (k + 1) % slots, (k - 1) % slots, 1]
        assert cw[12 + (k & 1)] == k + 1 and cw[14] == 0 and cw[15] == 0


@pytest.mark.parametrize("images", [True, False])
def test_actor_tail_in_one_launch_equals_the_two_launches(gpu_device, images):
    """tt_mlp_actor_tail: Q(s, mu(s)) / dQ/da through the updated critic and the actor's weight gradients + Adam + soft update + image
    patches in ONE grid -- dQ/da handed from the row workgroups to the weight-gradient workgroups in device memory (per row ONE
    agent-scope atomic word {learn step, value}) -- against the same two launches apart: every weight, both targets, the Adam
    moments, the flat gradients and q / dQ/da after 4 learn() calls on fixture F5's batch, bit for bit; also as a replayed hipGraph;
    the give-up word stays clear; a step counter set back (resume) does not match stale epoch words."""
    import torch
    from conftest import GOLDEN
    from ddpg_trucktrailer_amd.fused_learn import FusedLearner
    from test_learner import _batch
    z = np.load(os.path.join(GOLDEN, "f5_learner.npz"), allow_pickle=False)
    s, a, r, s2, d = _batch(z, gpu_device)
    d8 = d.to(torch.uint8)
    outs = []
    for tail, graph in ((False, False), (True, False), (True, True)):
        agent = _agent(gpu_device, z)
        fl = FusedLearner(agent, 256, fc2_images=images)
        fl.fuse_tail = tail
        fl.learn_batch(s, a, r, s2, d8)
        if graph:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=side):
                fl.learn_batch(s, a, r, s2, d8)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(3):
                g.replay()
        else:
            for _ in range(3):
                fl.learn_batch(s, a, r, s2, d8)
        torch.cuda.synchronize()
        assert fl.tail_gave_up() == 0 and int(fl.step_dev.item()) == 4
        if tail:
            tw = fl.tail_words.cpu()
            assert tw[:16].tolist() == [4] * 16 and tw[16:64].eq(-1).all()               # the row workgroups' hints
            rows = tw[64:64 + 512].view(256, 2)                                            # per row {float bits of dQ/da, learn step}
            assert rows[:, 1].eq(4).all() and torch.equal(rows[:, 0].contiguous().view(torch.float32), fl.dq_da.cpu())
            assert tw[64 + 512:].eq(-1).all()
        flat = torch.cat([p.detach().reshape(-1) for net in agent._nets() for p in net.parameters()])
        outs.append((flat.clone(), fl.actor.m.clone(), fl.actor.v.clone(), fl.critic.m.clone(), fl.actor.flat_grad.clone(),
                     fl.q_pi.clone(), fl.dq_da.clone()))
        if tail and not graph:      # resume to an earlier step: the words are cleared, the next learn() waits for ITS producers
            sd = fl.state_dict()
            sd["step"] = 3
            fl.load_state_dict(sd)
            assert fl.tail_words.eq(-1).all()
            fl.learn_batch(s, a, r, s2, d8)
            torch.cuda.synchronize()
            assert fl.tail_gave_up() == 0 and int(fl.step_dev.item()) == 4 and torch.isfinite(fl.dq_da).all()
    for k in (1, 2):
        for x, y in zip(outs[0], outs[k]):
            assert torch.equal(x, y), ("eager" if k == 1 else "graph")
    assert torch.isfinite(outs[0][0]).all()
