"""set_seed(seed): the host-side generators the single-env reference-API path draws from (DDPG/seed_utils.py:5-33) --
python's `random`, numpy's legacy global stream (env.reset(seed), OU noise, replay sampling) and torch (network init).
The N-env loop does not use them: its streams are counter-based Philox keyed by the loop's seed and device counters."""
import random

import numpy as np
import torch


def set_seed(seed, verbose=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if verbose:
        print(f"seed set to {seed} (python random, numpy, torch{', torch.cuda' if torch.cuda.is_available() else ''})")
    return seed
