import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irn_amd import ops
from oracle import irn_oracle as O
cm = np.load("tests/golden/cam_merge.npz")
dev = torch.device("cuda", 0)
for name in "ab":
    outs = [cm["%s_out%d" % (name, i)] for i in range(4)]
    size = tuple(int(v) for v in cm[name + "_size"])
    keys, cam, hi = ops.cam_merge([torch.from_numpy(o).to(dev) for o in outs], size, torch.from_numpy(cm[name + "_label"]))
    ok, olo, ohi = O.cam_merge(outs, size, cm[name + "_label"])
    for tag, a, b in (("lo", cam.cpu().numpy(), olo), ("hi", hi.cpu().numpy(), ohi)):
        d = np.abs(a - b)
        idx = np.unravel_index(d.argmax(), d.shape)
        print(name, tag, a.shape, "max diff %.3e at %s: %r vs %r; n differing %d; max per channel a %s b %s" % (d.max(), idx, a[idx], b[idx], (a != b).sum(), a.max(axis=(1, 2)), b.max(axis=(1, 2))))
    print([o.shape for o in outs], size)
