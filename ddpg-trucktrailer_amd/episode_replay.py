"""Episode record / replay data: the schema of DDPG/episode_replay_collector.py:13-26
`{states, actions, episode_num, env_data, info}` for chosen lanes of the vector env, kept on the device while
the episode runs and written as .npz (plain arrays, no pickle) when it ends."""
import os

import numpy as np
import torch

from ddpg_trucktrailer_amd import _lib as L


class EpisodeRecorder:
    """Records the episodes of `lanes` of a TruckTrailerVecEnv.

    call begin() after reset/set_pose, record(action, done, info) after every env.step(..., info=True);
    finished episodes are returned (and saved to `save_dir` as episode_<num>_reward_<int>.npz, the reference's
    file naming: episode_replay_collector.py:11-14)."""

    def __init__(self, env, lanes, save_dir=None):
        self.env, self.save_dir = env, save_dir
        self.lanes = torch.as_tensor(lanes, dtype=torch.long, device=env.device)
        self.episode_num = 0
        self._open = {}

    def begin(self):
        ep = self.env.episode()
        st = self.env.state[self.lanes].cpu().numpy()
        for j, lane in enumerate(self.lanes.tolist()):
            self._open[lane] = dict(
                states=[st[j]], actions=[], info=[],
                env_data={'startx': float(ep["start"][lane, 0]), 'starty': float(ep["start"][lane, 1]),
                          'startyaw': float(ep["start"][lane, 2]), 'goalx': float(ep["goal"][lane, 0]),
                          'goaly': float(ep["goal"][lane, 1]), 'goalyaw': float(ep["goal"][lane, 2])})

    def record(self, action, done, info, pre_reset_state=None):
        """action [N] f32 radians as given to env.step; info = the dict env.step(info=True) returned."""
        st = (self.env.state if pre_reset_state is None else pre_reset_state)[self.lanes].cpu().numpy()
        a = action[self.lanes].cpu().numpy()
        d = done[self.lanes].cpu().numpy().astype(bool)
        comp = info["comp"][:, self.lanes].cpu().numpy()
        viol = info["violation"][self.lanes].cpu().numpy()
        flags = info["flags"][self.lanes].cpu().numpy()
        finished = []
        for j, lane in enumerate(self.lanes.tolist()):
            ep = self._open.get(lane)
            if ep is None:
                continue
            ep["states"].append(st[j])
            ep["actions"].append(np.array([a[j]], np.float32))
            row = dict(zip(L.INFO_ROWS, comp[:, j]))
            row.update(violation_type=L.VIOLATIONS[int(viol[j])], success=bool(flags[j] & L.F_SUCCESS), distance_reward=0.0)
            ep["info"].append(row)
            if d[j]:
                finished.append(self._finish(lane))
        return finished

    def _finish(self, lane):
        ep = self._open.pop(lane)
        ep["episode_num"] = self.episode_num
        self.episode_num += 1
        if self.save_dir:
            os.makedirs(self.save_dir, exist_ok=True)
            total = sum(r["total_reward"] for r in ep["info"])
            save_episode(os.path.join(self.save_dir, f"episode_{ep['episode_num']}_reward_{int(total)}.npz"), ep)
        return ep


def save_episode(path, ep):
    keys = list(L.INFO_ROWS)
    np.savez_compressed(
        path, states=np.array(ep["states"], np.float64), actions=np.array(ep["actions"], np.float32),
        episode_num=np.int64(ep["episode_num"]),
        env_data=np.array([ep["env_data"][k] for k in ("startx", "starty", "startyaw", "goalx", "goaly", "goalyaw")]),
        info=np.array([[r[k] for k in keys] for r in ep["info"]], np.float64), info_keys=np.array(keys),
        violation=np.array([L.VIOLATIONS.index(r["violation_type"]) for r in ep["info"]], np.int32),
        success=np.array([r["success"] for r in ep["info"]], np.bool_))


def load_episode(path):
    z = np.load(path, allow_pickle=False)
    keys = [str(k) for k in z["info_keys"]]
    info = []
    for row, v, s in zip(z["info"], z["violation"], z["success"]):
        d = dict(zip(keys, (float(x) for x in row)))
        d.update(violation_type=L.VIOLATIONS[int(v)], success=bool(s))
        info.append(d)
    return {"states": list(z["states"]), "actions": list(z["actions"]), "episode_num": int(z["episode_num"]),
            "env_data": dict(zip(("startx", "starty", "startyaw", "goalx", "goaly", "goalyaw"), z["env_data"].tolist())),
            "info": info}
