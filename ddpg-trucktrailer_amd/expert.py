"""Expert-transition ingestion: trainv2.py:457-466 re-inserts stored transitions
`(obs, action, reward, obs_next, done)` one `agent.remember` call at a time; exp_gen.py:77-110 produces them with
the action already divided by radians(45).  Here they are loaded in bulk into the device replay memory:

  * `load_into_replay(buffer, ...)`  -- the per-transition `ReplayBuffer` the reference-style `Agent.learn()` samples;
  * `load_into_ring(ring, ...)`      -- the side buffer of the `TrajectoryRing` the N-env loop samples: expert tuples are
    not steps of any env's trajectory, so they sit next to the time-major ring and `tt_ring_sample` draws uniformly over
    ring transitions and side transitions together (the reference draws uniformly over one buffer that holds both)."""
import numpy as np


def transitions_to_arrays(stored_transitions):
    """list of episodes, each a list of (obs, action, reward, obs_next, done) -> five arrays."""
    rows = [t for episode in stored_transitions for t in episode]
    obs = np.stack([np.asarray(t[0], np.float32) for t in rows])
    act = np.stack([np.asarray(t[1], np.float32).reshape(-1) for t in rows])
    rew = np.array([float(t[2]) for t in rows], np.float32)
    obs2 = np.stack([np.asarray(t[3], np.float32) for t in rows])
    done = np.array([bool(t[4]) for t in rows], np.bool_)
    return obs, act, rew, obs2, done


def load_into_replay(buffer, stored_transitions):
    """Bulk version of the `agent.remember` loop of trainv2.py:462-465; returns the number loaded."""
    obs, act, rew, obs2, done = transitions_to_arrays(stored_transitions)
    buffer.store_batch(obs, act, rew, obs2, done)
    return len(rew)


def load_into_ring(ring, stored_transitions, capacity=None):
    """The same transitions into the N-env loop's replay (TrajectoryRing.load_side); returns the number loaded.
    Loops holding captured hipGraphs of the sampling launch re-capture afterwards (ring.side_epoch moves)."""
    obs, act, rew, obs2, done = transitions_to_arrays(stored_transitions)
    return ring.load_side(obs, act[:, 0], rew, obs2, done, capacity=capacity)
