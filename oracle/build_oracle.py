"""Builds the C oracle (test infrastructure).  `python oracle/build_oracle.py [--native]`.

Default flags are portable (the .so built in the build container travels to the GPU box);
--native adds -march=native and is what bench.py's cpu_baseline leg uses when it rebuilds the oracle
on the box it times it on."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(native=False, out_dir=None):
    out_dir = out_dir or os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libwalk_oracle%s.so" % ("_native" if native else ""))
    cmd = ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", out, os.path.join(HERE, "walk_oracle.c"), "-lm"]
    if native:
        cmd.insert(2, "-march=native")
    subprocess.check_call(cmd)
    return out


def load(native=False, out_dir=None):
    import ctypes as C
    path = os.path.join(out_dir or os.path.join(HERE, "_build"), "libwalk_oracle%s.so" % ("_native" if native else ""))
    if not os.path.exists(path):
        path = build(native, out_dir)
    lib = C.CDLL(path)
    lib.irn_oracle_walk.restype = C.c_int
    lib.irn_oracle_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.c_int, C.c_void_p]
    lib.irn_oracle_threads.restype = C.c_int
    lib.irn_oracle_walk_batch.restype = C.c_int
    lib.irn_oracle_walk_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_double, C.c_int, C.c_void_p]
    return lib


def walk(lib, cam, edge, radius, beta, n_sweeps):
    import numpy as np
    cam = np.ascontiguousarray(cam, np.float32)
    c = int(np.prod(cam.shape[:-2]))
    h, w = cam.shape[-2:]
    edge = np.ascontiguousarray(np.asarray(edge, np.float32).reshape(h, w))
    out = np.empty((c, 1, h, w), np.float32)
    rc = lib.irn_oracle_walk(edge.ctypes.data, cam.ctypes.data, c, h, w, int(radius), float(beta), int(n_sweeps),
                             out.ctypes.data)
    if rc:
        raise MemoryError("irn_oracle_walk failed")
    return out


def walk_batch(lib, cams, edges, radius, beta, n_sweeps):
    """The image-parallel row-vectorised form (the CPU baseline of bench.py): lists of cams [C_i,h_i,w_i] and edges
    [h_i,w_i] -> list of float32 [C_i,1,h_i,w_i]; one image per OpenMP thread."""
    import ctypes as C
    import numpy as np
    n = len(cams)
    cams = [np.ascontiguousarray(np.asarray(c, np.float32).reshape((-1,) + c.shape[-2:])) for c in cams]
    edges = [np.ascontiguousarray(np.asarray(e, np.float32).reshape(c.shape[-2:])) for e, c in zip(edges, cams)]
    outs = [np.empty((c.shape[0], 1) + c.shape[-2:], np.float32) for c in cams]
    ptrs = lambda arrs: (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    ints = lambda vals: (C.c_int * n)(*[int(v) for v in vals])
    rc = lib.irn_oracle_walk_batch(n, ptrs(edges), ptrs(cams), ints([c.shape[0] for c in cams]),
                                   ints([c.shape[1] for c in cams]), ints([c.shape[2] for c in cams]), int(radius),
                                   float(beta), int(n_sweeps), ptrs(outs))
    if rc:
        raise MemoryError("irn_oracle_walk_batch: %d images failed" % rc)
    return outs


if __name__ == "__main__":
    print(build(native="--native" in sys.argv))
