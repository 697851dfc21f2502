import numpy as np, sys
a, b = np.load("/tmp/a1.npz"), np.load("/tmp/b1.npz")
names = ["w1","b1","g1","be1","w2","b2","g2","be2","w3","b3","wa","ba"]
sizes = [9200,400,400,400,120000,300,300,300,300,1,300,300]
for net, nt in (("critic", 12), ("actor", 10)):
    g0, g1 = a[f"0/{net}/grad"], b[f"0/{net}/grad"]
    off = 0
    for n, s in zip(names[:nt], sizes[:nt]):
        d = np.abs(g0[off:off+s] - g1[off:off+s])
        if d.max() > 0:
            print(net, n, "differs in", int((d > 0).sum()), "of", s, "max", d.max(), "rel", d.max() / np.abs(g0[off:off+s]).max())
        off += s
