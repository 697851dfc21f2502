#!/usr/bin/env python3
"""GPU: the split-f16 forward (csrc/ttnet_split.hip) against the exact-f32 kernel, torch and torch-in-f64, its time, and
the corner cases of the two-piece f16 split (tiny weights/activations whose `m` piece is subnormal, large ones near the
stated range)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddpg_trucktrailer_amd import fused
from ddpg_trucktrailer_amd.networks import ActorNetwork, CriticNetwork

dev = torch.device("cuda:0")


def nets(seed=0, wscale=1.0):
    torch.manual_seed(seed)
    a = ActorNetwork(1e-4, (23,), 400, 300, 1, name="actor", device=dev)
    c = CriticNetwork(1e-3, (23,), 400, 300, 1, name="critic", device=dev)
    with torch.no_grad():
        for net in (a, c):
            net.bn1.weight.uniform_(0.5, 1.5); net.bn1.bias.uniform_(-0.3, 0.3)
            net.bn2.weight.uniform_(0.5, 1.5); net.bn2.bias.uniform_(-0.3, 0.3)
            net.fc2.weight.mul_(wscale); net.fc1.weight.mul_(wscale)
        a.mu.weight.uniform_(-0.2, 0.2); c.q.weight.uniform_(-0.2, 0.2)
    return a, c


def both(net, fn):
    w = fused.weights_of(net)
    ws = w.split_ws
    out_split = fn().clone()
    w.split_ws = None
    out_f32 = fn().clone()
    w.split_ws = ws
    return out_split, out_f32


def report(a, c, tag, ns=(1024, 1061, 5000, 65536)):
    for n in ns:
        obs = torch.rand((n, 23), device=dev) * 2 - 1
        act = torch.rand((n, 1), device=dev) * 2.4 - 1.2
        with torch.no_grad():
            ref_mu, ref_q = a(obs), c(obs, act)
            ref_mu64 = a.double()(obs.double()).float(); a.float()
            ref_q64 = c.double()(obs.double(), act.double()).float(); c.float()
        s, f = both(a, lambda: fused.actor_forward(a, obs))
        print(f"{tag} n={n} actor : |split-f32| {(s-f).abs().max().item():.2e}  vs f64: split {(s-ref_mu64).abs().max().item():.2e} "
              f"f32-kernel {(f-ref_mu64).abs().max().item():.2e} torch {(ref_mu-ref_mu64).abs().max().item():.2e}  nan {int(torch.isnan(s).sum())}")
        s, f = both(c, lambda: fused.critic_forward(c, obs, act))
        print(f"{tag} n={n} critic: |split-f32| {(s-f).abs().max().item():.2e}  vs f64: split {(s-ref_q64).abs().max().item():.2e} "
              f"f32-kernel {(f-ref_q64).abs().max().item():.2e} torch {(ref_q-ref_q64).abs().max().item():.2e}  (|q| max {ref_q.abs().max().item():.2f})")


a, c = nets(0)
report(a, c, "init-like")
a2, c2 = nets(1, wscale=20.0)          # |w| up to ~1.2 (trained-network sizes)
report(a2, c2, "w x20   ", ns=(5000,))
a3, c3 = nets(2, wscale=1e-3)          # tiny weights: m pieces subnormal
report(a3, c3, "w x1e-3 ", ns=(5000,))

for n in (65536, 262144):
    obs = torch.rand((n, 23), device=dev); out = torch.empty(n, device=dev)
    w = fused.weights_of(a); ws = w.split_ws
    for name in ("split", "f32"):
        w.split_ws = ws if name == "split" else None
        for _ in range(5): fused.actor_forward(a, obs, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fused.actor_forward(a, obs, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"N={n} {name}: {ms*1e3:.1f} us  ({n*2*(23*400+400*300+300)/ms/1e9:.1f} TFLOP/s f32-equivalent)")
    w.split_ws = ws
