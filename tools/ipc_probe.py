#!/usr/bin/env python3
"""Gate for the peer-to-peer gradient exchange: can a SECOND process open, on the same device, a
hipIpcMemHandle of memory that the first one allocated -- and do stores of one process reach the other while both run?

    python tools/ipc_probe.py            -> one JSON line {"ipc": true/false, ...}, exit code 0 either way

The parent never touches the GPU: it starts an owner and a peer process (this file again with a role argument) and relays
the 64-byte handle between them through pipes.  owner: hipMalloc 1 MiB, fill it, hipIpcGetMemHandle, print the handle; then
wait for the peer's answer and read back what the peer wrote.  peer: hipIpcOpenMemHandle, read, write a pattern into the second
half, close.  No torch tensors involved: plain HIP through ctypes (the runtime copy that torch loads, so that both
processes use the same one as the product)."""
import ctypes as C
import json
import os
import subprocess
import sys

N = 1 << 18      # floats


class Handle(C.Structure):      # hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to hipIpcOpenMemHandle
    _fields_ = [("reserved", C.c_ubyte * 64)]


def hip():
    import torch  # noqa: F401  (brings torch/lib/libamdhip64.so; the soname below then resolves to that copy)
    lib = C.CDLL("libamdhip64.so")
    lib.hipGetErrorString.restype = C.c_char_p
    lib.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
    lib.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    return lib


def ck(lib, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc} ({lib.hipGetErrorString(rc).decode()})")


def allow_any_tracer():
    """prctl(PR_SET_PTRACER, PR_SET_PTRACER_ANY): with the kernel's yama ptrace_scope = 1 only an ANCESTOR may pidfd_getfd() a
    process's file descriptors -- which is how the dmabuf form of a hipIpcMemHandle reaches the opener -- unless the owner
    names who else may.  Ranks of one job are siblings."""
    libc = C.CDLL(None, use_errno=True)
    return libc.prctl(0x59616D61, C.c_ulong(-1 & (2 ** 64 - 1)), 0, 0, 0)


def owner():
    import numpy as np
    if os.environ.get("TT_IPC_PTRACER") == "any":
        print("PRCTL", allow_any_tracer(), file=sys.stderr, flush=True)
    lib = hip()
    ck(lib, lib.hipSetDevice(0), "hipSetDevice")
    p = C.c_void_p()
    ck(lib, lib.hipMalloc(C.byref(p), C.c_size_t(4 * N)), "hipMalloc")
    src = np.arange(N, dtype=np.float32)
    ck(lib, lib.hipMemcpy(p, src.ctypes.data_as(C.c_void_p), C.c_size_t(4 * N), 1), "hipMemcpy H2D")
    handle = Handle()
    ck(lib, lib.hipIpcGetMemHandle(C.byref(handle), p), "hipIpcGetMemHandle")
    print("HANDLE " + bytes(handle.reserved).hex(), flush=True)
    line = sys.stdin.readline().strip()          # the peer is done
    out = np.zeros(N, dtype=np.float32)
    ck(lib, lib.hipDeviceSynchronize(), "hipDeviceSynchronize")
    ck(lib, lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(4 * N), 2), "hipMemcpy D2H")
    ok = bool((out[:N // 2] == src[:N // 2]).all() and (out[N // 2:] == -src[N // 2:]).all())
    print("OWNER " + json.dumps({"peer_said": line, "sees_peer_writes": ok}), flush=True)
    lib.hipFree(p)


def peer():
    import numpy as np
    lib = hip()
    ck(lib, lib.hipSetDevice(0), "hipSetDevice")
    raw = bytes.fromhex(sys.stdin.readline().strip())
    handle = Handle.from_buffer_copy(raw)
    p = C.c_void_p()
    rc = lib.hipIpcOpenMemHandle(C.byref(p), handle, 1)      # hipIpcMemLazyEnablePeerAccess
    if rc != 0:
        print("PEER " + json.dumps({"opened": False, "error": f"{rc} {lib.hipGetErrorString(rc).decode()}"}), flush=True)
        return
    got = np.zeros(N, dtype=np.float32)
    ck(lib, lib.hipMemcpy(got.ctypes.data_as(C.c_void_p), p, C.c_size_t(4 * N), 2), "hipMemcpy D2H")
    reads = bool((got == np.arange(N, dtype=np.float32)).all())
    neg = (-np.arange(N // 2, N, dtype=np.float32))
    ck(lib, lib.hipMemcpy(C.c_void_p(p.value + 4 * (N // 2)), neg.ctypes.data_as(C.c_void_p), C.c_size_t(4 * (N // 2)), 1), "hipMemcpy H2D")
    ck(lib, lib.hipDeviceSynchronize(), "hipDeviceSynchronize")
    ck(lib, lib.hipIpcCloseMemHandle(p), "hipIpcCloseMemHandle")
    print("PEER " + json.dumps({"opened": True, "reads_owner_data": reads}), flush=True)


def main(ptracer=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ptracer:
        env["TT_IPC_PTRACER"] = ptracer
    me = [sys.executable, os.path.abspath(__file__)]
    kw = dict(stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    res = {"ipc": False}
    a = subprocess.Popen(me + ["owner"], **kw)
    b = None
    try:
        line = a.stdout.readline()
        if not line.startswith("HANDLE "):
            res["error"] = "owner: " + (line + a.stderr.read())[-600:]
            return res
        b = subprocess.Popen(me + ["peer"], **kw)
        out, err = b.communicate(line.split()[1] + "\n", timeout=240)
        said = [x for x in out.splitlines() if x.startswith("PEER ")]
        if not said:
            res["error"] = "peer: " + (out + err)[-600:]
            a.stdin.write("failed\n"); a.stdin.flush()
            return res
        res["peer"] = json.loads(said[0][5:])
        a.stdin.write("done\n"); a.stdin.flush()
        out, err = a.communicate(timeout=120)
        said = [x for x in out.splitlines() if x.startswith("OWNER ")]
        res["owner"] = json.loads(said[0][6:]) if said else {"error": (out + err)[-600:]}
        res["ipc"] = bool(res["peer"].get("opened") and res["peer"].get("reads_owner_data") and res["owner"].get("sees_peer_writes"))
        return res
    finally:
        for p in (a, b):
            if p is not None and p.poll() is None:
                p.kill()


if __name__ == "__main__":
    role = sys.argv[1] if len(sys.argv) > 1 else "parent"
    if role == "owner":
        owner()
    elif role == "peer":
        peer()
    else:
        out = {"siblings": main()}
        try:
            out["yama_ptrace_scope"] = open("/proc/sys/kernel/yama/ptrace_scope").read().strip()
        except OSError as exc:
            out["yama_ptrace_scope"] = repr(exc)
        if not out["siblings"]["ipc"]:
            out["siblings_owner_allows_any_tracer"] = main("any")
        out["ipc"] = bool(out["siblings"]["ipc"] or out.get("siblings_owner_allows_any_tracer", {}).get("ipc"))
        print(json.dumps(out), flush=True)
