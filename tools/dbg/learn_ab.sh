#!/bin/bash
for img in ${IMGS:-1 0}; do
TT_LIB_PATH=$PWD/tools/dbg/libttenv_head.so python3 tools/dbg/learn_dump.py /tmp/a$img.npz $img || exit 1
python3 tools/dbg/learn_dump.py /tmp/b$img.npz $img || exit 1
python3 - <<PY
import numpy as np
a, b = np.load("/tmp/a$img.npz"), np.load("/tmp/b$img.npz")
print("images=$img")
for k in a.files:
    d = np.abs(a[k] - b[k]).max()
    if d > 0: print("  ", k, "max diff", d, "scale", np.abs(a[k]).max(), "first index", int(np.argmax(np.abs(a[k]-b[k]).reshape(-1))))
PY
done
