"""Driver entry points: build() compiles every HIP source for gfx950 (and the CPU oracle, which is
test infrastructure); smoke() runs one small invocation of the hot path on cuda:0 and checks it
against the oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "irn_amd", "csrc"), "-j", str(os.cpu_count() or 4)], env=env)
    lib = os.path.join(ROOT, "irn_amd", "lib", "libirn_hip.so")
    if not os.path.exists(lib):
        raise RuntimeError("build did not produce %s" % lib)
    # the checker: building it is not using it
    from oracle import build_oracle
    build_oracle.build(native=False)
    # a Python reference has nothing to compile into oracle/_ref: the reference is
    # run in the build container to write tests/golden/*.npz instead (tests/golden/make_golden.py)
    import irn_amd._lib as _lib          # loads the .so, checks every declared symbol resolves
    import irn_amd.misc.indexing         # noqa: F401
    import irn_amd.ops                   # noqa: F401
    import irn_amd.step.make_cam, irn_amd.step.make_sem_seg_labels, irn_amd.step.make_ins_seg_labels  # noqa: F401,E401
    assert _lib.lib.irn_version() >= 100


def smoke():
    import numpy as np
    import torch
    from irn_amd import ops, synth
    from irn_amd.misc import indexing
    from oracle import irn_oracle as O

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda", 0)
    h, w, c, r = 48, 56, 3, 5
    edge = synth.edge_field(h, w, seed=1)
    cam = synth.cam_blobs(c, h, w, seed=1)
    rw = indexing.propagate_to_edge(torch.from_numpy(cam).to(dev), torch.from_numpy(edge)[None].to(dev),
                                    radius=r, beta=10, exp_times=8)
    keys = np.array([2, 7, 14])
    out = ops.label_epilogue([rw], [(190, 221)], 0.25, keys=[torch.from_numpy(keys).to(dev)])
    torch.cuda.synchronize()
    st = O.propagate_to_edge_stencil(cam, edge, r, 10, 8)
    err = float(np.abs(rw.cpu().numpy() - st).max())
    assert err <= 1e-5, "walk differs from the fp64 oracle by %g" % err
    _, lab, _ = O.sem_seg_epilogue(rw.cpu().numpy(), (190, 221), keys, 0.25)
    assert np.array_equal(out["labels"][0].cpu().numpy(), lab), "labels differ from the oracle"
    assert "libirn_hip.so" in open("/proc/self/maps").read()
    print("smoke ok: walk max-abs vs fp64 oracle %.2e, labels identical (%d px)" % (err, lab.size))


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
