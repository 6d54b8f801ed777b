#!/usr/bin/env python3
"""Derive a find database whose channels-last picks accumulate in a fixed order, from the tuned one shipped with the package.

    python tools/miopen_det_filter.py [--src irn_amd/data/miopen/<key>] [--dst irn_amd/data/miopen/<key>-det]

MIOpen's deterministic attribute rules out every fast NHWC fp32 solver wholesale (profiles/r05_s3_deterministic_ab.txt), although
only ONE kind of kernel in the tuned channels-last trunk is order-dependent: `ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC` with a
tuned configuration that splits K across workgroups (`gemm_k_global_split` != 0: partial sums meet in atomics).  This tool copies
the database and, for every NHWC fp32 forward problem whose record lists that solver with such a configuration, removes the
solver from the record — MIOpen's fast find then takes the next entry (the composable-kernel grouped convolution, 3-14 % slower
on those layers, no split-K) — and leaves everything else alone.  `irn_amd/step/_common.miopen_setup` seeds from `<key>-det` in
the reproducible mode; whether the result IS bit-stable is measured, not assumed (tools/determinism_probe.py, tests).
Reference: the convolutions of net/resnet50.py:17-108 as the steps batch them."""
import argparse
import glob
import os
import shutil
import sys

GTC = "ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC"
GKS_FIELD = 17          # fwd,nhwc,fp32 + 14 tile parameters, then gemm_k_global_split


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irn_amd", "data", "miopen")
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default=None)
    ap.add_argument("--dst", default=None)
    a = ap.parse_args()
    src = a.src or sorted(d for d in glob.glob(os.path.join(root, "*")) if os.path.isdir(d) and not d.endswith("-det"))[0]
    dst = a.dst or src.rstrip("/") + "-det"
    os.makedirs(dst, exist_ok=True)
    udb = glob.glob(os.path.join(src, "*.udb.txt"))[0]
    ufdb = glob.glob(os.path.join(src, "*.ufdb.txt"))[0]
    # problems whose tuned GTC configuration splits K: the perf database is keyed differently from the find database
    # (C x H x W x ... vs C-H-W-...), so the decision is made per perf-db line and matched through the shared fields
    split = {}
    for line in open(udb):
        if "=" not in line:
            continue
        key, val = line.rstrip("\n").split("=", 1)
        for ent in val.split(";"):
            if ent.startswith(GTC + ":"):
                f = ent.split(":", 1)[1].split(",")
                split[key] = len(f) > GKS_FIELD and f[GKS_FIELD] != "0"

    def find_sig(fkey):
        # find-db key  C-H-W-KhxKw-K-Ho-Wo-N-PhxPw-ShxSw-DhxDw-0-<layouts>-FP32-F
        p = fkey.split("-")
        return (p[0], p[1], p[2]) + tuple(p[3].split("x")) + (p[4], p[7]) + tuple(p[8].split("x")) + tuple(p[9].split("x")) + tuple(p[10].split("x"))

    def perf_sig(ukey):
        # perf-db key  2xCxHxWx1xKhxKwx1xKxNxPhxPwx0xShxSwx0xDhxDwx0x0x1x<layout>xFP32xF
        p = ukey.split("x")
        return (p[1], p[2], p[3], p[5], p[6], p[8], p[9], p[10], p[11], p[13], p[14], p[16], p[17])

    split_sig = {perf_sig(k): v for k, v in split.items() if "NHWC" in k}
    removed = kept = 0
    out = []
    for line in open(ufdb):
        if "=" not in line:
            out.append(line)
            continue
        key, val = line.rstrip("\n").split("=", 1)
        ents = val.split(";")
        if "NHWC" in key and key.endswith("-F") and any(e.startswith(GTC + ":") for e in ents):
            if split_sig.get(find_sig(key), True) and len(ents) > 2:       # unknown configuration: treated as splitting
                ents = [e for e in ents if not e.startswith(GTC + ":")]
                removed += 1
            else:
                kept += 1
        out.append(key + "=" + ";".join(ents) + "\n")
    for f in os.listdir(src):
        if os.path.isfile(os.path.join(src, f)):
            shutil.copy2(os.path.join(src, f), os.path.join(dst, f))
    with open(os.path.join(dst, os.path.basename(ufdb)), "w") as fh:
        fh.writelines(out)
    print("%s -> %s: GTC implicit GEMM removed from %d channels-last forward records (split-K configuration), kept in %d" % (src, dst, removed, kept))


if __name__ == "__main__":
    main()
