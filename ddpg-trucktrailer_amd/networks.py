"""Actor / critic of the reference DDPG agent on PyTorch-ROCm (DDPG/networks.py:9-174).

Same module and parameter names as the reference (fc1, fc2, bn1, bn2, mu / action_value, q), so its
state_dicts load here and ours load there; same custom uniform init ranges; Adam with the reference's
settings (critic: weight_decay=0.01 folded into the gradient, i.e. torch's Adam, not AdamW)."""
import os

import numpy as np
import torch as T
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim


def _pick_device(device):
    if device is not None:
        return T.device(device)
    return T.device('cuda:0' if T.cuda.is_available() else 'cpu')


class _Checkpointed(nn.Module):
    """save/load helpers shared by both nets (networks.py:70-95, 149-174)."""

    def _init_paths(self, name, chkpt_dir):
        self.name = name
        self.checkpoint_dir = chkpt_dir
        self.checkpoint_file = os.path.join(self.checkpoint_dir, name + '_ddpg')

    def save_checkpoint(self):
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        T.save(self.state_dict(), self.checkpoint_file)

    def save_checkpoint_progress(self, success):
        d = os.path.join(self.checkpoint_dir, str(success))
        os.makedirs(d, exist_ok=True)
        T.save(self.state_dict(), os.path.join(d, self.name + '_ddpg'))

    def load_checkpoint(self):
        self.load_state_dict(T.load(self.checkpoint_file, map_location=self.device, weights_only=True))

    def save_best(self):
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        T.save(self.state_dict(), os.path.join(self.checkpoint_dir, self.name + '_best'))


def _uniform_(layer, bound):
    layer.weight.data.uniform_(-bound, bound)
    layer.bias.data.uniform_(-bound, bound)


class CriticNetwork(_Checkpointed):
    """Q(s, a): Linear(23,400) -> LayerNorm -> ReLU -> Linear(400,300) -> LayerNorm, + Linear(1,300)(a),
    ReLU(sum), Linear(300,1)  (networks.py:21-68)."""

    def __init__(self, beta, input_dims, fc1_dims, fc2_dims, n_actions, name, chkpt_dir='tmp/ddpg', device=None,
                 capturable=False):
        super().__init__()
        self.input_dims, self.fc1_dims, self.fc2_dims, self.n_actions = input_dims, fc1_dims, fc2_dims, n_actions
        self._init_paths(name, chkpt_dir)
        self.fc1 = nn.Linear(*self.input_dims, self.fc1_dims)
        self.fc2 = nn.Linear(self.fc1_dims, self.fc2_dims)
        self.bn1 = nn.LayerNorm(self.fc1_dims)
        self.bn2 = nn.LayerNorm(self.fc2_dims)
        self.action_value = nn.Linear(self.n_actions, self.fc2_dims)
        self.q = nn.Linear(self.fc2_dims, 1)
        # init ranges: 1/sqrt(out_features) for fc1, fc2 and the action layer, 0.003 for q (networks.py:33-47)
        _uniform_(self.fc1, 1. / np.sqrt(self.fc1.weight.data.size()[0]))
        _uniform_(self.fc2, 1. / np.sqrt(self.fc2.weight.data.size()[0]))
        _uniform_(self.q, 0.003)
        _uniform_(self.action_value, 1. / np.sqrt(self.action_value.weight.data.size()[0]))
        self.device = _pick_device(device)
        self.to(self.device)
        on_gpu = self.device.type == 'cuda'   # fused = one multi-tensor kernel per step; same update rule
        self.optimizer = optim.Adam(self.parameters(), lr=beta, weight_decay=0.01, capturable=capturable and on_gpu,
                                    fused=on_gpu)

    def forward(self, state, action):
        state_value = F.relu(self.bn1(self.fc1(state)))
        state_value = self.bn2(self.fc2(state_value))
        action_value = self.action_value(action)
        return self.q(F.relu(T.add(state_value, action_value)))


class ActorNetwork(_Checkpointed):
    """mu(s): Linear(23,400) -> LayerNorm -> ReLU -> Linear(400,300) -> LayerNorm -> ReLU -> Linear(300,1) -> tanh
    (networks.py:110-147)."""

    def __init__(self, alpha, input_dims, fc1_dims, fc2_dims, n_actions, name, chkpt_dir='tmp/ddpg', device=None,
                 capturable=False):
        super().__init__()
        self.input_dims, self.fc1_dims, self.fc2_dims, self.n_actions = input_dims, fc1_dims, fc2_dims, n_actions
        self._init_paths(name, chkpt_dir)
        self.fc1 = nn.Linear(*self.input_dims, self.fc1_dims)
        self.fc2 = nn.Linear(self.fc1_dims, self.fc2_dims)
        self.bn1 = nn.LayerNorm(self.fc1_dims)
        self.bn2 = nn.LayerNorm(self.fc2_dims)
        self.mu = nn.Linear(self.fc2_dims, self.n_actions)
        # same draw order as the reference: fc2 first, then fc1, then mu (networks.py:121-131)
        _uniform_(self.fc2, 1. / np.sqrt(self.fc2.weight.data.size()[0]))
        _uniform_(self.fc1, 1. / np.sqrt(self.fc1.weight.data.size()[0]))
        _uniform_(self.mu, 0.003)
        self.device = _pick_device(device)
        self.to(self.device)
        on_gpu = self.device.type == 'cuda'
        self.optimizer = optim.Adam(self.parameters(), lr=alpha, capturable=capturable and on_gpu, fused=on_gpu)

    def forward(self, state):
        x = F.relu(self.bn1(self.fc1(state)))
        x = F.relu(self.bn2(self.fc2(x)))
        return T.tanh(self.mu(x))
