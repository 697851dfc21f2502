"""Can this node replay an RCCL all-reduce from inside a hipGraph?  Asked ONCE per data-parallel run, in throw-away
child processes, before the ranks themselves touch a GPU.

The data-parallel vector step is one hipGraph when its two gradient all-reduces can be captured with the rest
(rollout.DDPGRollout(graph_collectives=True)); otherwise three graph segments with eager collectives between them.  A
capture that the collective library mishandles does not raise -- it hangs or corrupts -- so the answer is not found by
trying it in the process that must go on.  Each rank starts `python -m ddpg_trucktrailer_amd.dp_probe` (its own process
group on MASTER_PORT + 1, one child per rank), which captures an AVG all-reduce on a side stream, replays it and checks the
numbers; the rank waits with a time limit, ends exactly the process group it started when the limit passes, and takes
"no" for an answer on any failure.  Ranks need not agree: a graph-replayed collective and an eager one are the same call
sequence to RCCL."""
import os
import signal
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def graph_collectives_ok(timeout=240.0):
    """Run in a rank that has NOT initialised a GPU.  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment."""
    if int(os.environ.get("WORLD_SIZE", "1")) < 2 and os.environ.get("TT_DP_PROBE_FORCE") != "1":
        return False
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1)
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
