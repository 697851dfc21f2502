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


def probe_port(timeout=30.0, tag="collective"):
    """The port the probe's children rendezvous on: chosen free by rank 0, read by the others from the launcher's store (one
    port per `tag`: the two probes of a run do not share one).  Touches no GPU.  Fallback (no store reachable): MASTER_PORT + 1
    (+ 2 for the exchange's probe)."""
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
        key = "tt/dp_probe_port/" + tag
        if rank == 0:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            store.set(key, str(port))
        return int(store.get(key).decode())
    except Exception:
        return master + (2 if tag == "p2p" else 1)


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
    return _ask("collective", "graph-collective-ok", timeout, port)


def p2p_exchange_ok(timeout=180.0, port=None):
    """This rank's VOTE on the peer-to-peer gradient exchange (include/ttenv.h: tt_p2p_*): a child process per rank opens the peers'
    exchange blocks through IPC handles (gloo carries them), runs the exchange's optimizer launch for both sites over several steps
    -- eagerly and as a replayed hipGraph -- and checks that every rank ends with the same bits and with the mean of the ranks'
    gradients applied.  Whatever fails or does not finish within the limit means "no"; the caller reduces the votes with agree()."""
    return _ask("p2p", "p2p-exchange-ok", timeout, port)


def _ask(what, marker, timeout, port):
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
        proc = subprocess.Popen([sys.executable, "-m", "ddpg_trucktrailer_amd.dp_probe", what], env=env, cwd=_ROOT,
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
    if os.environ.get("TT_DP_PROBE_VERBOSE") == "1" and not (proc.returncode == 0 and marker in (out or "")):
        print(f"dp_probe[{what}] rank {os.environ.get('RANK', '0')}: rc {proc.returncode}\n{(out or '')[-1500:]}", file=sys.stderr, flush=True)
    return proc.returncode == 0 and marker in (out or "")


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


def _child_p2p():
    """One rank of the exchange's probe (a throw-away process).  Control plane: gloo (it only carries 128 bytes per rank)."""
    import ctypes as C
    from datetime import timedelta
    import torch
    import torch.distributed as dist
    from ddpg_trucktrailer_amd import _lib as L
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank))) % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    lib = L.load()
    sizes = [132201, 131601]                            # the two sites' flat gradients
    h = C.c_void_p()
    L.check_p2p(lib.tt_p2p_create(local, rank, world, 2, (C.c_int32 * 2)(*sizes), C.byref(h)))
    L.check_p2p(lib.tt_p2p_set_timeout(h, 20.0), h)
    mine = C.create_string_buffer(L.P2P_HANDLE_BYTES)
    L.check_p2p(lib.tt_p2p_export(h, mine), h)
    handles = [None] * world
    dist.all_gather_object(handles, bytes(mine.raw))
    for r, hb in enumerate(handles):
        if r != rank:
            L.check_p2p(lib.tt_p2p_attach(h, r, C.create_string_buffer(hb, L.P2P_HANDLE_BYTES)), h)
    dist.barrier()
    from ddpg_trucktrailer_amd.fused_learn import _device_view
    f = dict(dtype=torch.float32, device=dev)
    grads = [_device_view(lib.tt_p2p_grad(h, s), n, dev) for s, n in enumerate(sizes)]
    st = [dict(p=torch.zeros(n, **f), m=torch.zeros(n, **f), v=torch.zeros(n, **f)) for n in sizes]
    ref = [dict(p=torch.zeros(n, **f), m=torch.zeros(n, **f), v=torch.zeros(n, **f), g=torch.zeros(n, **f)) for n in sizes]
    step_dev = torch.zeros((), dtype=torch.int64, device=dev)
    ptr = lambda t: (C.c_void_p * 1)(t.data_ptr())
    stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def fill(site, step):            # this rank's "gradient": different on every rank, different every step
        i = torch.arange(sizes[site], **f)
        grads[site].copy_(torch.sin(i * 0.37 + step) * (rank + 1) + 0.01 * site)

    def exchange(site):
        n = sizes[site]
        L.check_p2p(lib.tt_adam_soft_update_p2p(h, site, 1, ptr(st[site]["p"]), ptr(st[site]["m"]), ptr(st[site]["v"]), None,
                                                (C.c_int32 * 1)(n), C.c_void_p(step_dev.data_ptr()), 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.0,
                                                None, None, stream()), h)

    def reference(site, step):       # the same update from the mean formed in rank order, by the launch without an exchange
        i = torch.arange(sizes[site], **f)
        acc = torch.zeros(sizes[site], **f)
        for r in range(world):
            acc = acc + (torch.sin(i * 0.37 + step) * (r + 1) + 0.01 * site)
        ref[site]["g"].copy_(acc / world if world > 1 else acc)
        L.check(lib.tt_adam_soft_update(1, ptr(ref[site]["p"]), ptr(ref[site]["g"]), ptr(ref[site]["m"]), ptr(ref[site]["v"]), None,
                                        (C.c_int32 * 1)(sizes[site]), C.c_void_p(step_dev.data_ptr()), 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.0,
                                        None, None, stream()))

    graph = None
    for step in range(1, 7):         # steps 1-3 eager, 4-6 as replays of one captured graph of both sites' launches
        step_dev.fill_(step)
        fill(0, step); fill(1, step)
        torch.cuda.synchronize()
        dist.barrier()               # (the ranks enter an exchange together: its wait is bounded)
        if step < 4:
            exchange(0); exchange(1)
        else:
            if graph is None:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream())
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    exchange(0); exchange(1)
                torch.cuda.current_stream().wait_stream(side)
            graph.replay()
        reference(0, step); reference(1, step)
        torch.cuda.synchronize()
        assert int(lib.tt_p2p_gave_up(h)) == 0, "an exchange wait was abandoned"
        for site in (0, 1):
            got, want = st[site]["p"], ref[site]["p"]
            assert torch.isfinite(got).all() and float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), (step, site)
            all_p = [torch.empty_like(got, device="cpu") for _ in range(world)]
            dist.all_gather(all_p, got.cpu())
            assert all(torch.equal(all_p[0], x) for x in all_p[1:]), f"ranks differ at step {step}, site {site}"
    dist.barrier()
    dist.destroy_process_group()
    print("p2p-exchange-ok", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "p2p":
        _child_p2p()
    else:
        _child()
