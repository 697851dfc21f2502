"""Can this node replay an RCCL all-reduce from inside a hipGraph?  Asked ONCE per data-parallel run, in throw-away
child processes, before the ranks themselves touch a GPU.

The data-parallel vector step is one hipGraph when its two gradient all-reduces can be captured with the rest
(rollout.DDPGRollout(graph_collectives=True)); otherwise three graph segments with eager collectives between them.  A
capture that the collective library mishandles does not raise -- it hangs or corrupts -- so the answer is not found by
trying it in the process that must go on.  Each rank starts `python -m ddpg_trucktrailer_amd.dp_probe` (its own process
group on MASTER_PORT + 1, one child per rank), which captures an AVG all-reduce on a side stream, replays it and checks the
numbers; the rank waits with a time limit, ends exactly the process group it started when the limit passes, and takes
"no" for an answer on any failure.

The ranks of a run must all build the SAME launch structure (a rank replaying 20-step graphs with RCCL nodes inside beside
a rank running three segments with eager all-reduces has never run anywhere), so a rank's own answer is only a vote:
after init_process_group the caller reduces the votes with agree() (all-reduce MIN) and passes the result to
DDPGRollout(graph_collectives=...) explicitly.  The probe's rendezvous port is not guessed: probe_port() has rank 0 bind
a free port and hand it to the others through the launcher's store (torch.distributed.run's agent store on MASTER_PORT);
where no such store answers, MASTER_PORT + 1 is the fallback -- a collision there makes some rank vote "no", and the
reduced answer is then "no" on every rank."""
import os
import signal
import socket
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def probe_port(timeout=30.0):
    """The port the probe's children rendezvous on: chosen free by rank 0, read by the others from the launcher's store.
    Touches no GPU.  Fallback (no store reachable): MASTER_PORT + 1."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    master = int(os.environ.get("MASTER_PORT", "29500"))
    if os.environ.get("TT_DP_PROBE_PORT"):
        return int(os.environ["TT_DP_PROBE_PORT"])
    if world < 2:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]
    try:
        from datetime import timedelta
        import torch.distributed as dist
        store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), master, world, is_master=False,
                              timeout=timedelta(seconds=timeout), wait_for_workers=False)
        key = "tt/dp_probe_port"
        if rank == 0:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            store.set(key, str(port))
        return int(store.get(key).decode())
    except Exception:
        return master + 1


def agree(vote, device=None):
    """The run's answer from the ranks' votes: all-reduce MIN over the default process group (after init_process_group).
    Every rank returns the same bool."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return bool(vote)
    t = torch.tensor([1 if vote else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def graph_collectives_ok(timeout=240.0, port=None):
    """This rank's VOTE (see agree()).  Run in a rank that has NOT initialised a GPU.  Reads RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_* from the environment; port: the probe's rendezvous port (default: probe_port())."""
    if int(os.environ.get("WORLD_SIZE", "1")) < 2 and os.environ.get("TT_DP_PROBE_FORCE") != "1":
        return False
    env = dict(os.environ)
    env["MASTER_PORT"] = str(port if port is not None else probe_port())
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = _ROOT + os.pathsep + env.get("PYTHONPATH", "")
    # the elastic launcher's variables describe the PARENT's rendezvous; the child gets a plain env:// one
    for k in [k for k in env if k.startswith("TORCHELASTIC_") or k.startswith("TORCH_NCCL_ASYNC")]:
        env.pop(k)
    try:
        proc = subprocess.Popen([sys.executable, "-m", "ddpg_trucktrailer_amd.dp_probe"], env=env, cwd=_ROOT,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    except OSError:
        return False
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)      # the session this call started, nothing else
        except OSError:
            pass
        proc.wait()
        return False
    return proc.returncode == 0 and "graph-collective-ok" in (out or "")


def _child():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank))) % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from datetime import timedelta
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timedelta(seconds=120))
    n = 132201                                          # the critic's flat gradient
    x = torch.full((n,), float(rank + 1), device=dev)
    dist.all_reduce(x, op=dist.ReduceOp.AVG)            # communicator set up outside any capture
    torch.cuda.synchronize()
    want = sum(range(1, world + 1)) / world
    assert float(x[0]) == want and float(x[-1]) == want
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    y = torch.empty_like(x)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        y.mul_(2.0)
        dist.all_reduce(y, op=dist.ReduceOp.AVG)
        y.add_(1.0)
    torch.cuda.current_stream().wait_stream(side)
    for it in range(3):
        y.fill_(float(rank + 1 + it))
        g.replay()
        torch.cuda.synchronize()
        w = 2.0 * (sum(range(1, world + 1)) / world + it) + 1.0
        assert float(y[0]) == w and float(y[n // 2]) == w and float(y[-1]) == w, (it, float(y[0]), w)
    dist.barrier()
    dist.destroy_process_group()
    print("graph-collective-ok", flush=True)


if __name__ == "__main__":
    _child()
