"""Replay memory.

ReplayBuffer mirrors DDPG/replay_buffer.py:3-34 (store_transition / sample_buffer, uniform sampling WITH
replacement through numpy's global RNG) with the storage on the device in float32 instead of float64 host
arrays.  TrajectoryRing is the vector form used by the N-env loop: the env kernel writes straight into it."""
import numpy as np
import torch


class ReplayBuffer:
    def __init__(self, max_size, input_shape, n_actions, device=None):
        self.mem_size = int(max_size)
        self.mem_cntr = 0
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        f32 = dict(dtype=torch.float32, device=self.device)
        self.state_memory = torch.zeros((self.mem_size, *input_shape), **f32)
        self.new_state_memory = torch.zeros((self.mem_size, *input_shape), **f32)
        self.action_memory = torch.zeros((self.mem_size, n_actions), **f32)
        self.reward_memory = torch.zeros(self.mem_size, **f32)
        self.terminal_memory = torch.zeros(self.mem_size, dtype=torch.bool, device=self.device)

    def store_transition(self, state, action, reward, state_, done):
        index = self.mem_cntr % self.mem_size
        self.state_memory[index] = torch.as_tensor(np.asarray(state), dtype=torch.float32)
        self.action_memory[index] = torch.as_tensor(np.asarray(action), dtype=torch.float32)
        self.reward_memory[index] = float(reward)
        self.new_state_memory[index] = torch.as_tensor(np.asarray(state_), dtype=torch.float32)
        self.terminal_memory[index] = bool(done)
        self.mem_cntr += 1

    def store_batch(self, states, actions, rewards, states_, dones):
        """Bulk insert (expert transitions, trainv2.py:457-466): k transitions at once, wrapping around."""
        k = len(rewards)
        idx = (torch.arange(k, device=self.device) + self.mem_cntr) % self.mem_size
        self.state_memory[idx] = torch.as_tensor(states, dtype=torch.float32, device=self.device)
        self.action_memory[idx] = torch.as_tensor(actions, dtype=torch.float32, device=self.device).reshape(k, -1)
        self.reward_memory[idx] = torch.as_tensor(rewards, dtype=torch.float32, device=self.device)
        self.new_state_memory[idx] = torch.as_tensor(states_, dtype=torch.float32, device=self.device)
        self.terminal_memory[idx] = torch.as_tensor(dones, dtype=torch.bool, device=self.device)
        self.mem_cntr += k

    def sample_buffer(self, batch_size):
        max_mem = min(self.mem_cntr, self.mem_size)
        batch = torch.as_tensor(np.random.choice(max_mem, batch_size), device=self.device)
        return (self.state_memory[batch], self.action_memory[batch], self.reward_memory[batch],
                self.new_state_memory[batch], self.terminal_memory[batch])


class TrajectoryRing:
    """Time-major ring of the last T vector steps of N envs, all on the device, f32.

    obs[t] is the observation the policy saw at vector step t and obs[t+1] the one the env returned, so a
    transition (t, n) is (obs[t][n], act[t][n], rew[t][n], obs[t+1][n], done[t][n]) and each observation is
    stored ONCE (101 B per transition instead of 193 B).  For a finished env obs[t+1][n] is the first
    observation of its next episode; DDPG zeroes Q' for done transitions (DDPG_agent.py:89), so that row is
    never used.  The env kernel writes obs[t+1], rew[t], done[t] in place: inserting costs no copy."""

    def __init__(self, n_envs, slots, obs_dim, device):
        assert slots >= 3
        self.n, self.slots, self.device = n_envs, slots, device
        self.obs = torch.zeros((slots, n_envs, obs_dim), dtype=torch.float32, device=device)
        self.act = torch.zeros((slots, n_envs), dtype=torch.float32, device=device)
        self.rew = torch.zeros((slots, n_envs), dtype=torch.float32, device=device)
        self.done = torch.zeros((slots, n_envs), dtype=torch.uint8, device=device)
        self.k = 0                                                          # vector steps completed (host)
        self.k_dev = torch.zeros((), dtype=torch.int64, device=device)      # same, on the device (graph-safe)
        self._env_counts = False                                            # True: the env's step kernel advances k_dev
        # stand-alone transitions (expert tuples, trainv2.py:457-466) that take part in every uniform draw.  Fixed
        # capacity and fixed addresses (a captured sampling launch keeps pointing at them); only `count` moves, and a
        # launch bakes it by value: holders of captured graphs re-capture when side_epoch moves
        self.side = None
        self.side_count = 0
        self.side_epoch = 0
        # {t, t+1, t-1, t > 0}: ring slots of the running vector step, written on the device by the step's opening launch
        # (include/ttenv.h: tt_ring_view / tt_ring_cursor) so that captured launches need no per-position pointers
        # [4..7] / [8..11]: the cursors {t, t+1, t-1, t > 0} of even / odd steps, [0..3]: the running step's (include/ttenv.h)
        # ring cursors [0..11] + the image hand-over words [12..15] (include/ttenv.h: TT_CURSOR_INTS)
        # [16]: the step chain's progress (k + 1 once the policy launch of step k has begun): what a pipelined learn() waits for
        # [18..19]: address of the host word below (TT_CURSOR_GAVE_UP_MIRROR)
        self.cursor_dev = torch.zeros(32, dtype=torch.int32, device=device)
        # a launch that gives up waiting for the other chain marks cursor[15] AND this word of pinned (device-visible) host
        # memory: the host sees a give-up by reading its own memory, with no copy or synchronize in the loop
        self.gave_up_host = None
        if torch.device(device).type == "cuda":
            self.gave_up_host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._gave_up_np = self.gave_up_host.numpy()
            self._write_mirror_address()

    def _write_mirror_address(self):
        if self.gave_up_host is not None:
            self.cursor_dev[18:20] = torch.tensor([self.gave_up_host.data_ptr()], dtype=torch.int64).view(torch.int32).to(self.device)

    def attach(self, env):
        """Let the env's step kernel advance k_dev (tt_env_set_step_counter): one launch less per vector step.  From
        then on every env.step()/step_random() of that env counts as one stored vector step."""
        env.set_step_counter(self.k_dev)
        self._env_counts = True

    @property
    def capacity(self):
        return (self.slots - 1) * self.n

    def __len__(self):
        return min(self.k, self.slots - 1) * self.n

    def slot(self, k=None):
        return (self.k if k is None else k) % self.slots

    def advance(self):
        self.k += 1
        if not self._env_counts:
            self.k_dev += 1

    # ---- expert / stand-alone transitions ---------------------------------------------------------------
    def load_side(self, obs, act, rew, obs2, done, capacity=None):
        """Append k stand-alone transitions (obs [k,23], act [k] or [k,1], rew [k], obs2 [k,23], done [k]) to the side
        buffer the sampler mixes into its uniform draw: the bulk form of the reference's `agent.remember` loop over
        stored expert transitions (trainv2.py:457-466).  The first call fixes the capacity (default: k)."""
        f = dict(dtype=torch.float32, device=self.device)
        obs = torch.as_tensor(obs, **f).reshape(-1, self.obs.shape[2])
        k = obs.shape[0]
        if self.side is None:
            cap = int(capacity) if capacity is not None else k
            d = self.obs.shape[2]
            self.side = dict(obs=torch.zeros((cap, d), **f), act=torch.zeros(cap, **f), rew=torch.zeros(cap, **f),
                             obs2=torch.zeros((cap, d), **f), done=torch.zeros(cap, dtype=torch.uint8, device=self.device))
        cap = self.side["act"].shape[0]
        if self.side_count + k > cap:
            raise ValueError(f"side buffer holds {cap} transitions, {self.side_count} used, {k} more do not fit")
        a, b = self.side_count, self.side_count + k
        self.side["obs"][a:b] = obs
        self.side["act"][a:b] = torch.as_tensor(act, **f).reshape(-1)
        self.side["rew"][a:b] = torch.as_tensor(rew, **f).reshape(-1)
        self.side["obs2"][a:b] = torch.as_tensor(obs2, **f).reshape(-1, self.obs.shape[2])
        self.side["done"][a:b] = torch.as_tensor(np.asarray(done).astype(np.uint8) if not torch.is_tensor(done) else done,
                                                 device=self.device).to(torch.uint8).reshape(-1)
        self.side_count = b
        self.side_epoch += 1
        return k

    def _side_struct(self):
        from ddpg_trucktrailer_amd import _lib as L
        if self.side is None or self.side_count == 0:
            return None
        sd = self.side
        return L.TTSideBuffer(sd["obs"].data_ptr(), sd["act"].data_ptr(), sd["rew"].data_ptr(), sd["obs2"].data_ptr(),
                              sd["done"].data_ptr(), int(self.side_count), 0)

    # ---- checkpoint ----------------------------------------------------------------------------------------
    def state_dict(self, with_replay=True):
        sd = {"k": int(self.k), "slots": int(self.slots), "n": int(self.n), "side_count": int(self.side_count)}
        if with_replay:
            sd.update(obs=self.obs.cpu(), act=self.act.cpu(), rew=self.rew.cpu(), done=self.done.cpu())
            if self.side is not None:
                sd["side"] = {k: v.cpu() for k, v in self.side.items()}
        return sd

    def load_state_dict(self, sd):
        assert (int(sd["slots"]), int(sd["n"])) == (self.slots, self.n), "replay ring geometry differs"
        if "obs" in sd:
            self.obs.copy_(sd["obs"]); self.act.copy_(sd["act"]); self.rew.copy_(sd["rew"]); self.done.copy_(sd["done"])
            if "side" in sd:
                cap = sd["side"]["act"].shape[0]
                if self.side is None or self.side["act"].shape[0] != cap:
                    self.side = {k: v.to(self.device).clone() for k, v in sd["side"].items()}
                else:
                    for k, v in sd["side"].items():
                        self.side[k].copy_(v)
        # the side buffer's count always follows the checkpoint: a file without side tuples (or saved with_replay=False)
        # must not leave this ring's older tuples in the draw; captured launches bake the count, so the epoch moves with it
        count = int(sd.get("side_count", 0)) if ("obs" in sd and "side" in sd) else 0
        if count != self.side_count:
            self.side_count = count
            self.side_epoch += 1
        self.k = int(sd["k"])               # the counters come back whether or not the contents did
        self.k_dev.fill_(self.k)
        self.cursor_dev.zero_()             # image epochs are step numbers: a counter set back must not find newer ones
        self.cursor_dev[16] = self.k        # ... and the step chain is where the counters say (TT_CURSOR_PROGRESS)
        self._write_mirror_address()
        if self.gave_up_host is not None:
            self._gave_up_np[0] = 0

    def policy_gave_up(self):
        """Step number + 1 at which a launch stopped waiting for the other chain -- a policy launch for its image, or learn()'s first
        launch for the step chain's progress (include/ttenv.h: TT_CURSOR_GAVE_UP, TT_CURSOR_PROGRESS) --, 0 = never.
        Synchronises."""
        return int(self.cursor_dev[15].item()) or self.gave_up_seen()

    def gave_up_seen(self):
        """The same mark as the host sees it WITHOUT touching the GPU (the launch that gives up also sets a word of pinned host
        memory: TT_CURSOR_GAVE_UP_MIRROR): what a loop looks at between graph replays.  A mark not yet visible here is seen
        by the next look; policy_gave_up() is the exact, synchronising form."""
        return int(self._gave_up_np[0]) if self.gave_up_host is not None else 0

    def mark_gave_up_for_test(self, step):
        """What a launch of vector step `step` that gives up leaves behind (device word and host mirror): tests of the fallback."""
        self.cursor_dev[15] = int(step) + 1
        torch.cuda.current_stream(self.device).synchronize()
        if self.gave_up_host is not None:
            self._gave_up_np[0] = int(step) + 1

    def clear_gave_up(self):
        self.cursor_dev[15] = 0
        if self.gave_up_host is not None:
            torch.cuda.current_stream(self.device).synchronize()
            self._gave_up_np[0] = 0

    def view(self):
        from ddpg_trucktrailer_amd import _lib as L
        return L.TTRingView(self.cursor_dev.data_ptr(), self.obs.data_ptr(), self.act.data_ptr(), self.rew.data_ptr(),
                            self.done.data_ptr(), self.n, self.slots)

    def cursor(self, counter=None):
        """tt_ring_cursor for a step's opening launch; counter: the device step counter to take the step number from (default:
        the ring's own; a loop whose opening launch may run before the previous env step has advanced that one passes its
        own count of opened steps)."""
        from ddpg_trucktrailer_amd import _lib as L
        return L.TTRingCursor((self.k_dev if counter is None else counter).data_ptr(), self.slots, 0, self.cursor_dev.data_ptr())

    def _batch_bufs(self, batch_size):
        # one set per batch size, kept for the ring's lifetime: captured graphs hold these addresses, so a draw of another
        # size must not free (and hand to someone else) the buffers a graph still writes
        cache = self.__dict__.setdefault("_buf_cache", {})
        if batch_size not in cache:
            f = dict(dtype=torch.float32, device=self.device)
            d = self.obs.shape[2]
            cache[batch_size] = (torch.empty((batch_size, d), **f), torch.empty((batch_size, 1), **f), torch.empty(batch_size, **f),
                                 torch.empty((batch_size, d), **f), torch.empty(batch_size, dtype=torch.uint8, device=self.device),
                                 torch.empty((batch_size, 2), dtype=torch.int32, device=self.device))
        self._bufs = cache[batch_size]          # (the set of the latest draw)
        return self._bufs

    def sample_args(self, batch_size, seed=0, k_dev=None, reserve=0, lag=0, draws=1, seed_stride=0, wait_for_steps=False):
        """tt_sample_args for one draw into the ring's batch buffers (include/ttenv.h); keeps what it points at alive.
        draws > 1 (tt_mlp_split_pack_and_sample only): that many draws of batch_size rows, draw u with seed + u * seed_stride
        into rows [u * batch_size, (u + 1) * batch_size) of the buffers _batch_bufs(draws * batch_size).
        wait_for_steps (tt_mlp_forward_multi_sampled only): the launch first waits, in device memory, until the step chain has
        reached step *k_dev - 1 (cursor word 16, written by every ring-addressed policy launch): tt_sample_args.step_progress."""
        from ddpg_trucktrailer_amd import _lib as L
        s, a, r, s2, dn, idx = self._batch_bufs(batch_size * max(1, int(draws)))
        p = lambda t: t.data_ptr()
        side = self._side_struct()
        self._side_keep = side
        import ctypes as C
        return L.TTSampleArgs(batch_size, self.n, self.slots, int(reserve), p(self.k_dev if k_dev is None else k_dev),
                              p(self.obs), p(self.act), p(self.rew), p(self.done), int(seed) & (2 ** 64 - 1),
                              C.pointer(side) if side is not None else None, p(s), p(a), p(r), p(s2), p(dn), p(idx), int(lag),
                              max(1, int(draws)), int(seed_stride) & (2 ** 64 - 1),
                              self.cursor_dev.data_ptr() + 4 * 16 if wait_for_steps else None)

    def sample_fused(self, batch_size, seed=0, return_index=False, done_as_bool=True, k_dev=None, reserve=0, lag=0):
        """sample() as ONE HIP launch (tt_ring_sample): Philox indices keyed by (seed, k_dev) + gather.
        done_as_bool=False returns the raw uint8 flags (no conversion launch; what the fused learner takes).
        k_dev / reserve / lag: another device step counter, the number of newest slots to keep out of the window and the
        number of counted steps that may still be under way (a pipelined loop samples BESIDE the env steps of the running
        and of the previous vector step: include/ttenv.h, tt_ring_sample)."""
        import ctypes as C
        from ddpg_trucktrailer_amd import _lib as L
        s, a, r, s2, dn, idx = self._batch_bufs(batch_size)
        p = lambda t: C.c_void_p(t.data_ptr())
        side = self._side_struct()
        L.check(L.load().tt_ring_sample(batch_size, self.n, self.slots, p(self.k_dev if k_dev is None else k_dev), p(self.obs),
                                        p(self.act), p(self.rew), p(self.done), int(seed) & (2 ** 64 - 1), int(reserve), int(lag),
                                        C.byref(side) if side is not None else None,
                                        p(s), p(a), p(r), p(s2), p(dn), p(idx),
                                        C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        out = (s, a, r, s2, dn.bool() if done_as_bool else dn)
        return out + (idx,) if return_index else out

    def sample(self, batch_size, generator=None):
        """Uniform with replacement over the stored transitions; index math on the device (k_dev), so the call
        can sit inside a captured hipGraph."""
        dev = self.device
        u = torch.rand((2, batch_size), device=dev, generator=generator)
        avail = torch.clamp(self.k_dev, max=self.slots - 1)                          # complete transitions, in steps
        back = (u[0] * avail).long().clamp_(max=self.slots - 2)                      # 0 = newest
        t = torch.remainder(self.k_dev - 1 - back, self.slots)
        n = (u[1] * self.n).long().clamp_(max=self.n - 1)
        t1 = torch.remainder(t + 1, self.slots)
        out = [self.obs[t, n], self.act[t, n].unsqueeze(1), self.rew[t, n], self.obs[t1, n], self.done[t, n].bool()]
        if self.side is not None and self.side_count > 0:      # the same mixed uniform draw as tt_ring_sample
            m = self.side_count
            in_ring = (avail * self.n).to(torch.float64)
            pick = torch.rand(batch_size, device=dev, generator=generator, dtype=torch.float64) * (in_ring + m) < m
            j = (torch.rand(batch_size, device=dev, generator=generator) * m).long().clamp_(max=m - 1)
            sd = self.side
            for i, src in enumerate((sd["obs"][j], sd["act"][j].unsqueeze(1), sd["rew"][j], sd["obs2"][j], sd["done"][j].bool())):
                sel = pick.view(-1, *([1] * (src.dim() - 1)))
                out[i] = torch.where(sel, src, out[i])
        return tuple(out)
