"""GPU (-m gpu): the fused actor / critic inference kernels -- exact-f32 MFMA (csrc/ttnet.hip) and, from 1024 rows,
split-f16 MFMA (csrc/ttnet_split.hip) -- against the plain torch modules (f32 reference of the same op), tolerance
2e-5 absolute on tanh outputs / 2e-5 relative on Q; the two kernels against each other and against torch in f64."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nets(dev, seed=0):
    import torch
    from ddpg_trucktrailer_amd.networks import ActorNetwork, CriticNetwork
    torch.manual_seed(seed)
    a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
    c = CriticNetwork(1e-3, (23,), 400, 300, 1, name="critic", device=dev)
    with torch.no_grad():    # non-trivial LayerNorm affine + bigger heads so that mistakes show
        for net in (a, c):
            net.bn1.weight.uniform_(0.5, 1.5); net.bn1.bias.uniform_(-0.3, 0.3)
            net.bn2.weight.uniform_(0.5, 1.5); net.bn2.bias.uniform_(-0.3, 0.3)
        a.mu.weight.uniform_(-0.2, 0.2); c.q.weight.uniform_(-0.2, 0.2)
    return a, c


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 1024, 1153, 65536])
@pytest.mark.parametrize("kernel", ["default", "f32"])
def test_actor_and_critic_forward_match_torch(gpu_device, n, kernel):
    """default = what the launcher picks (exact f32 below 1024 rows, split-f16 from there); f32 = exact f32 forced."""
    import contextlib
    import torch
    from ddpg_trucktrailer_amd import fused
    actor, critic = _nets(gpu_device)
    assert fused.supported(actor) and fused.supported(critic)
    g = torch.Generator(device=gpu_device).manual_seed(n)
    obs = torch.rand((n, 23), device=gpu_device, generator=g) * 2 - 1
    act = torch.rand((n, 1), device=gpu_device, generator=g) * 2.4 - 1.2
    with torch.no_grad():
        ref_mu, ref_q = actor(obs), critic(obs, act)
    with (fused.exact_f32(actor) if kernel == "f32" else contextlib.nullcontext()), \
            (fused.exact_f32(critic) if kernel == "f32" else contextlib.nullcontext()):
        mu = fused.actor_forward(actor, obs)
        q = fused.critic_forward(critic, obs, act)
    assert mu.shape == ref_mu.shape and q.shape == ref_q.shape
    assert (mu - ref_mu).abs().max().item() <= 2e-5
    assert (q - ref_q).abs().max().item() <= 2e-5 * max(1.0, ref_q.abs().max().item())


def test_split_f16_kernel_is_f32_accurate(gpu_device):
    """The split-f16 kernel (two round-to-nearest f16 pieces per f32 operand, 3 of the 4 partial products) against the
    exact-f32 kernel and against the modules evaluated in f64: its error is of the size of f32 rounding itself -- also with
    trained-size weights (x20) and with weights so small that their second pieces are f16 subnormals (x1e-3)."""
    import torch
    from ddpg_trucktrailer_amd import fused
    actor, critic = _nets(gpu_device, seed=11)
    n = 20000 + 77                                     # not a multiple of the 128-row workgroup tile
    obs = torch.rand((n, 23), device=gpu_device) * 2 - 1
    act = torch.rand((n, 1), device=gpu_device) * 2.4 - 1.2
    mu_s, q_s = fused.actor_forward(actor, obs).clone(), fused.critic_forward(critic, obs, act).clone()
    with fused.exact_f32(actor), fused.exact_f32(critic):
        mu_e, q_e = fused.actor_forward(actor, obs).clone(), fused.critic_forward(critic, obs, act).clone()
    assert not torch.equal(mu_s, mu_e)                 # really two kernels
    assert (mu_s - mu_e).abs().max().item() <= 1e-5 and (q_s - q_e).abs().max().item() <= 2e-5
    with torch.no_grad():
        mu64 = actor.double()(obs.double()).float(); q64 = critic.double()(obs.double(), act.double()).float()
        actor.float(); critic.float()
        mu_t = actor(obs)
    err_s, err_e, err_t = ((x - mu64).abs().max().item() for x in (mu_s, mu_e, mu_t))
    assert err_s <= 3 * max(err_e, err_t) + 1e-7, (err_s, err_e, err_t)
    assert (q_s - q64).abs().max().item() <= 3 * (q_e - q64).abs().max().item() + 1e-6
    for scale in (20.0, 1e-3):
        with torch.no_grad():
            actor.fc2.weight.mul_(scale); actor.fc1.weight.mul_(scale)
            mu64 = actor.double()(obs.double()).float(); actor.float()
        mu_s = fused.actor_forward(actor, obs).clone()
        with fused.exact_f32(actor):
            mu_e = fused.actor_forward(actor, obs).clone()
        assert torch.isfinite(mu_s).all()
        assert (mu_s - mu64).abs().max().item() <= 3 * (mu_e - mu64).abs().max().item() + 2e-7, scale
        with torch.no_grad():
            actor.fc2.weight.div_(scale); actor.fc1.weight.div_(scale)


def test_real_observations_and_weight_updates_are_seen(gpu_device):
    """Env observations as input; an in-place optimizer-style update of the weights changes the output."""
    import torch
    from ddpg_trucktrailer_amd import fused
    from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
    actor, _ = _nets(gpu_device, seed=3)
    env = TruckTrailerVecEnv(5000)
    obs = env.reset(seed=1)
    with torch.no_grad():
        ref = actor(obs)
    assert (fused.actor_forward(actor, obs) - ref).abs().max().item() <= 2e-5
    with torch.no_grad():
        actor.fc2.weight.mul_(1.01); actor.mu.bias.add_(0.05)
        ref2 = actor(obs)
    out2 = fused.actor_forward(actor, obs)
    assert (out2 - ref2).abs().max().item() <= 2e-5 and (out2 - ref).abs().max().item() > 1e-3
    env.close()


def test_fused_choose_action_epilogue(gpu_device):
    import torch
    from ddpg_trucktrailer_amd import fused
    actor, _ = _nets(gpu_device, seed=5)
    n = 40000
    obs = torch.rand((n, 23), device=gpu_device) * 2 - 1
    ou = torch.zeros(n, device=gpu_device)
    raw, scaled, mu = (torch.empty(n, device=gpu_device) for _ in range(3))
    high = float(np.float32(math.pi / 4))
    x_prev = ou.clone()
    draws = []
    for step in range(6):
        fused.actor_act(actor, obs, ou, raw, scaled, seed=27, step=step, mu_out=mu, high=high)
        with torch.no_grad():
            assert (mu.view(-1, 1) - actor(obs)).abs().max().item() <= 2e-5
        assert torch.allclose(raw, mu + ou, atol=1e-7)                              # a = mu + noise (DDPG_agent.py:41-45)
        assert torch.equal(scaled, torch.clamp(raw, -1, 1) * high)                  # trainv2.py:516
        nrm = (ou - x_prev * (1 - 0.2 * 0.01)) / (0.15 * math.sqrt(0.01))           # the N(0,1) that was drawn
        draws.append(nrm)
        x_prev = ou.clone()
    z = torch.stack(draws)
    assert abs(z.mean().item()) < 0.01 and abs(z.std().item() - 1) < 0.01
    assert abs((z ** 4).mean().item() - 3) < 0.1                                    # Gaussian kurtosis
    assert abs(torch.corrcoef(torch.stack([draws[0], draws[1]]))[0, 1].item()) < 0.02   # steps independent
    # restart of the noise for finished envs; device-side step counter
    done = torch.zeros(n, dtype=torch.uint8, device=gpu_device); done[::2] = 1
    step_dev = torch.tensor(100, dtype=torch.int64, device=gpu_device)
    before = ou.clone()
    fused.actor_act(actor, obs, ou, raw, scaled, seed=27, step=0, step_dev=step_dev, done_prev=done, high=high)
    assert ((ou[::2]).abs() < 0.15 * 0.1 * 6).all() and not torch.equal(ou[1::2], before[1::2])
    again = before.clone()
    fused.actor_act(actor, obs, again, raw, scaled, seed=27, step=100, done_prev=done, high=high)
    assert torch.equal(again, ou)                                                    # step + *step_dev is the counter


def test_launch_geometry_does_not_change_results(gpu_device):
    """The split kernel's tiles in one grid, in capped grids (tt_mlp_weights.max_workgroups) and in a few capped grids
    followed by one whole-chip grid (capped_grids) are the same rows computed by the same code: bit-identical."""
    import torch
    from ddpg_trucktrailer_amd import fused
    actor, _ = _nets(gpu_device, seed=9)
    n = 128 * 37 + 55                                   # 38 tiles, the last one ragged
    obs = torch.rand((n, 23), device=gpu_device) * 2 - 1
    fused.pack(actor, 1)
    outs = []
    for wg, capped in ((0, 0), (8, 0), (8, 2), (5, 1), (64, 3)):
        w = fused.packed_weights_of(actor, 1, wg, capped)
        out = torch.full((n,), float("nan"), device=gpu_device)
        from ddpg_trucktrailer_amd import _lib as L
        import ctypes as C
        L.check(L.load().tt_actor_forward(n, C.c_void_p(obs.data_ptr()), C.byref(w), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)))
        outs.append(out.clone())
    fused.packed_weights_of(actor, 1, 0, 0)
    with torch.no_grad():
        assert (outs[0].view(-1, 1) - actor(obs)).abs().max().item() <= 2e-5
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
