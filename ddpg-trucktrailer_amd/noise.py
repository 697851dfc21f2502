"""Ornstein-Uhlenbeck exploration noise (DDPG/noise.py:3-20): theta 0.2, sigma 0.15, dt 1e-2, reset to 0.

OUActionNoise is the reference's single-env object (numpy global RNG, same call sequence, so the same
np.random.seed gives the same noise).  VecOUNoise keeps one OU state per env on the device."""
import numpy as np
import torch


class OUActionNoise:
    def __init__(self, mu, sigma=0.15, theta=0.2, dt=1e-2, x0=None):
        self.theta, self.mu, self.sigma, self.dt, self.x0 = theta, mu, sigma, dt, x0
        self.reset()

    def __call__(self):
        x = self.x_prev + self.theta * (self.mu - self.x_prev) * self.dt + \
            self.sigma * np.sqrt(self.dt) * np.random.normal(size=self.mu.shape)
        self.x_prev = x
        return x

    def reset(self):
        self.x_prev = self.x0 if self.x0 is not None else np.zeros_like(self.mu)


class VecOUNoise:
    """x <- x + theta*(mu - x)*dt + sigma*sqrt(dt)*N(0,1) for every env; envs whose episode ended restart at 0
    (trainv2.py:492 resets the noise at every episode start)."""

    def __init__(self, n_envs, device, mu=0.0, sigma=0.15, theta=0.2, dt=1e-2, generator=None):
        self.mu, self.sigma, self.theta, self.dt = mu, sigma, theta, dt
        self.x = torch.zeros(n_envs, dtype=torch.float32, device=device)
        self.generator = generator
        self._scale = float(sigma * np.sqrt(dt))

    def sample(self, normals=None):
        if normals is None:
            normals = torch.randn(self.x.shape, dtype=torch.float32, device=self.x.device, generator=self.generator)
        self.x.add_(self.theta * self.dt * (self.mu - self.x)).add_(normals, alpha=self._scale)
        return self.x

    def reset(self, done=None):
        if done is None:
            self.x.zero_()
        else:
            self.x.masked_fill_(done.bool(), 0.0)
