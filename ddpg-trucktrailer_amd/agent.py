"""DDPG agent with the reference's interface (DDPG/DDPG_agent.py:9-131) on PyTorch-ROCm.

    Agent(alpha, beta, input_dims, tau, n_actions, gamma, max_size, fc1_dims, fc2_dims, batch_size)
    .choose_action(observation, evaluate) .remember(s, a, r, s_, done) .learn()
    .save_models() .load_models() .save_models_progress(success) .update_network_parameters(tau)

learn() keeps the reference's order of operations (DDPG_agent.py:72-106): targets from the target nets,
critic MSE step, THEN the actor step through the already-updated critic, then the soft update of every
named parameter (LayerNorm included).  `learn_batch` is the same update on an explicit device batch; it is
what the N-env loop (rollout.py) captures into a hipGraph and where the data-parallel gradient all-reduce
sits (two sites: before critic.optimizer.step and before actor.optimizer.step)."""
import numpy as np
import torch as T
import torch.nn.functional as F

from ddpg_trucktrailer_amd.networks import ActorNetwork, CriticNetwork
from ddpg_trucktrailer_amd.noise import OUActionNoise
from ddpg_trucktrailer_amd.replay_buffer import ReplayBuffer


class GradAllReduce:
    """Flat-bucket gradient averaging over the data-parallel group (RCCL over xGMI on GPUs, gloo on CPU).
    One all-reduce per network per learn(): critic 132,201 f32 = 529 KB, actor 131,601 f32 = 526 KB."""

    def __init__(self, params, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.params = [p for p in params]
        self.world = dist.get_world_size(group)
        n = sum(p.numel() for p in self.params)
        self.flat = T.zeros(n, dtype=self.params[0].dtype, device=self.params[0].device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def __call__(self):
        T._foreach_copy_(self.views, [p.grad for p in self.params])
        self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        T._foreach_copy_([p.grad for p in self.params], self.views)


class Agent():
    def __init__(self, alpha, beta, input_dims, tau, n_actions, gamma=0.99,
                 max_size=1000000, fc1_dims=400, fc2_dims=300,
                 batch_size=64, device=None, chkpt_dir='tmp/ddpg', capturable=False, replay=True):
        self.gamma, self.tau, self.batch_size, self.alpha, self.beta = gamma, tau, batch_size, alpha, beta
        self.device = T.device(device) if device is not None else T.device('cuda:0' if T.cuda.is_available() else 'cpu')
        self.memory = ReplayBuffer(max_size, input_dims, n_actions, device=self.device) if replay else None
        self.noise = OUActionNoise(mu=np.zeros(n_actions))
        kw = dict(device=self.device, chkpt_dir=chkpt_dir, capturable=capturable)
        self.actor = ActorNetwork(alpha, input_dims, fc1_dims, fc2_dims, n_actions=n_actions, name='actor', **kw)
        self.critic = CriticNetwork(beta, input_dims, fc1_dims, fc2_dims, n_actions=n_actions, name='critic', **kw)
        self.target_actor = ActorNetwork(alpha, input_dims, fc1_dims, fc2_dims, n_actions=n_actions,
                                         name='target_actor', **kw)
        self.target_critic = CriticNetwork(beta, input_dims, fc1_dims, fc2_dims, n_actions=n_actions,
                                           name='target_critic', **kw)
        self.update_network_parameters(tau=1)
        self._critic_params = list(self.critic.parameters())
        self._actor_params = list(self.actor.parameters())
        self.grad_sync_actor = self.grad_sync_critic = None
        self.fused_targets = False
        self.fused_learner = None      # set by DDPGRollout when learn() runs on the hand-fused kernels
        self.last_critic_loss = self.last_actor_loss = None

    # ------------------------------------------------------------------ acting (DDPG_agent.py:36-49)
    def choose_action(self, observation, evaluate=False):
        self.actor.eval()
        state = T.as_tensor(np.asarray([observation]), dtype=T.float).to(self.device)
        with T.no_grad():
            mu = self.actor.forward(state)
        if not evaluate:
            mu = mu + T.tensor(self.noise(), dtype=T.float).to(self.device)
        self.actor.train()
        return mu.cpu().detach().numpy()[0]

    def remember(self, state, action, reward, state_, done):
        self.memory.store_transition(state, action, reward, state_, done)

    # ------------------------------------------------------------------ checkpoints (DDPG_agent.py:54-70)
    def _nets(self):
        return (self.actor, self.target_actor, self.critic, self.target_critic)

    def save_models(self):
        for net in self._nets():
            net.save_checkpoint()

    def save_models_progress(self, success):
        for net in self._nets():
            net.save_checkpoint_progress(success=success)

    def load_models(self):
        for net in self._nets():
            net.load_checkpoint()

    # ------------------------------------------------------------------ learning (DDPG_agent.py:72-106)
    def enable_data_parallel(self, group=None):
        """Average gradients over the process group at the two optimizer sites; broadcast rank 0's weights."""
        import torch.distributed as dist
        for net in self._nets():
            for p in net.parameters():
                dist.broadcast(p.data, src=0, group=group)
        self.grad_sync_critic = GradAllReduce(self.critic.parameters(), group)
        self.grad_sync_actor = GradAllReduce(self.actor.parameters(), group)

    def learn(self):
        if self.memory.mem_cntr < self.batch_size:
            return
        states, actions, rewards, states_, done = self.memory.sample_buffer(self.batch_size)
        self.learn_batch(states, actions, rewards, states_, done)

    def learn_batch(self, states, actions, rewards, states_, done):
        with T.no_grad():
            if self.fused_targets:      # one fused f32-MFMA launch per target net (csrc/ttnet.hip)
                from ddpg_trucktrailer_amd import fused
                target_actions = fused.actor_forward(self.target_actor, states_)
                critic_value_ = fused.critic_forward(self.target_critic, states_, target_actions)
            else:
                target_actions = self.target_actor.forward(states_)
                critic_value_ = self.target_critic.forward(states_, target_actions)
            critic_value_ = critic_value_.masked_fill(done.view(-1, 1), 0.0).view(-1)      # critic_value_[done] = 0.0
            target = (rewards + self.gamma * critic_value_).view(-1, 1)
        critic_value = self.critic.forward(states, actions)

        # Gradients through torch.autograd.grad, not .backward(): backward() delivers them through each parameter's
        # AccumulateGrad node, which is pinned to the stream it was first used on -- a node kept alive from another stream
        # (e.g. by a clone of the parameter that still carries its grad_fn) forks a hipGraph capture of this function and
        # HIP's EndCapture then crashes the process.  grad() returns the same numbers without touching those nodes.
        critic_loss = F.mse_loss(target, critic_value)
        for p, g in zip(self._critic_params, T.autograd.grad(critic_loss, self._critic_params)):
            p.grad = g                                       # zero_grad(set_to_none=True) + backward()
        if self.grad_sync_critic is not None:
            self.grad_sync_critic()
        self.critic.optimizer.step()

        # The actor step differentiates -Q(s, mu(s)) through the ALREADY UPDATED critic.  The reference lets
        # that backward also deposit gradients in the critic's parameters and throws them away at the next
        # zero_grad (DDPG_agent.py:95,100-104); asking for the actor's gradients only changes nothing the optimizers see.
        actor_loss = T.mean(-self.critic.forward(states, self.actor.forward(states)))
        for p, g in zip(self._actor_params, T.autograd.grad(actor_loss, self._actor_params)):
            p.grad = g
        if self.grad_sync_actor is not None:
            self.grad_sync_actor()
        self.actor.optimizer.step()

        self.update_network_parameters()
        self.last_critic_loss, self.last_actor_loss = critic_loss.detach(), actor_loss.detach()

    def update_network_parameters(self, tau=None):
        """theta' <- tau*theta + (1 - tau)*theta' over every named parameter (DDPG_agent.py:108-131)."""
        if tau is None:
            tau = self.tau
        with T.no_grad():
            for net, target in ((self.critic, self.target_critic), (self.actor, self.target_actor)):
                src = [p.data for p in net.parameters()]
                dst = [p.data for p in target.parameters()]
                if tau == 1:
                    T._foreach_copy_(dst, src)
                else:
                    T._foreach_lerp_(dst, src, tau)     # theta' + tau*(theta - theta'): one multi-tensor kernel
