"""Non-executing decoder for the reference's one recorded episode.

The reference ships `DDPG/episode_replays/episode_10579_reward_4792.pkl`
(format: DDPG/episode_replay_collector.py:15-21).  Serialized files from the
reference must never be unpickled, so this module does NOT use
`pickle.load`/`Unpickler`.  It walks the opcode stream with
`pickletools.genops` (a pure disassembler: it imports nothing and calls
nothing named by the file) and interprets the handful of opcodes that occur
with a symbolic stack machine:

* GLOBAL/STACK_GLOBAL push a *name* (`Sym`), never an imported object;
* REDUCE/BUILD record `(callee-name, args)` / state tuples as inert nodes;
* after the walk, nodes whose callee name is one of the three numpy
  reconstructors are turned into arrays/scalars with `np.frombuffer` on the
  raw bytes.  Any other name raises.

Only used by `make_golden.py` in the build container; the decoded episode is
re-encoded as `f1_golden_episode.npz` (plain arrays, allow_pickle=False).
"""
import pickletools

import numpy as np


class Sym:
    def __init__(self, module, name):
        self.module, self.name = module, name

    @property
    def full(self):
        return f"{self.module}.{self.name}"

    def __repr__(self):
        return f"Sym({self.full})"


class Node:
    """Inert record of `callee(*args)` plus optional BUILD state."""

    def __init__(self, callee, args):
        self.callee, self.args, self.state = callee, args, None


_MARK = object()

_ALLOWED = {
    "numpy._core.multiarray._reconstruct",
    "numpy.core.multiarray._reconstruct",
    "numpy._core.multiarray.scalar",
    "numpy.core.multiarray.scalar",
    "numpy.ndarray",
    "numpy.dtype",
}


def _walk(data: bytes):
    stack, memo = [], []

    def pop_mark():
        items = []
        while True:
            x = stack.pop()
            if x is _MARK:
                break
            items.append(x)
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "STOP":
            return stack.pop()
        elif n == "MARK":
            stack.append(_MARK)
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "MEMOIZE":
            memo.append(stack[-1])
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BININT", "BININT1", "BININT2",
                   "BINFLOAT", "SHORT_BINBYTES", "BINBYTES"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "TUPLE1":
            a = stack.pop()
            stack.append((a,))
        elif n == "TUPLE2":
            b, a = stack.pop(), stack.pop()
            stack.append((a, b))
        elif n == "TUPLE3":
            c, b, a = stack.pop(), stack.pop(), stack.pop()
            stack.append((a, b, c))
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n == "STACK_GLOBAL":
            name, module = stack.pop(), stack.pop()
            sym = Sym(module, name)
            if sym.full not in _ALLOWED:
                raise ValueError(f"unexpected global {sym.full}")
            stack.append(sym)
        elif n == "REDUCE":
            args, callee = stack.pop(), stack.pop()
            if not isinstance(callee, Sym):
                raise ValueError("REDUCE on a non-symbol")
            stack.append(Node(callee, args))
        elif n == "BUILD":
            state = stack.pop()
            node = stack[-1]
            if not isinstance(node, Node):
                raise ValueError("BUILD on a non-node")
            node.state = state
        elif n == "APPENDS":
            items = pop_mark()
            stack[-1].extend(items)
        elif n == "APPEND":
            x = stack.pop()
            stack[-1].append(x)
        elif n == "SETITEMS":
            items = pop_mark()
            d = stack[-1]
            for k, v in zip(items[0::2], items[1::2]):
                d[k] = v
        elif n == "SETITEM":
            v, k = stack.pop(), stack.pop()
            stack[-1][k] = v
        else:
            raise ValueError(f"opcode {n} not supported by the non-executing decoder")
    raise ValueError("no STOP")


def _dtype_of(node):
    if not (isinstance(node, Node) and node.callee.full == "numpy.dtype"):
        raise ValueError("expected a numpy.dtype node")
    code = node.args[0]
    if code not in ("f4", "f8", "b1", "i8", "i4"):
        raise ValueError(f"unexpected dtype code {code}")
    dt = np.dtype(code)
    if node.state is not None and node.state[1] in ("<", ">"):
        dt = dt.newbyteorder(node.state[1])
    return dt


def _materialize(x):
    if isinstance(x, Node):
        full = x.callee.full
        if full.endswith("multiarray._reconstruct"):
            _ver, shape, dt_node, fortran, raw = x.state
            arr = np.frombuffer(raw, dtype=_dtype_of(dt_node)).reshape(shape)
            if fortran:
                arr = arr.reshape(shape[::-1]).T
            return arr.copy()
        if full.endswith("multiarray.scalar"):
            dt_node, raw = x.args
            return np.frombuffer(raw, dtype=_dtype_of(dt_node))[0]
        raise ValueError(f"unexpected top-level node {full}")
    if isinstance(x, dict):
        return {k: _materialize(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_materialize(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_materialize(v) for v in x)
    if isinstance(x, Sym):
        raise ValueError("bare symbol in data")
    return x


def decode_episode(path):
    with open(path, "rb") as f:
        data = f.read()
    return _materialize(_walk(data))
