#!/usr/bin/env python3
"""Group a rocprofv3 kernel_stats.csv of a backbone-bound leg by what the kernels are (GEMM, Winograd, elementwise ...)."""
import collections
import csv
import sys


def kind(n):
    if n.startswith("Cijk"):
        return "rocBLAS/Tensile GEMM (1x1 conv)"
    if "miopenSp3AsmConv" in n:
        return "MIOpen Winograd (3x3 conv)"
    if "igemm" in n or "naive_conv" in n or "Conv" in n and "miopen" in n.lower():
        return "MIOpen other conv: " + n[:48]
    if "bn_act" in n:
        return "irn bn_act " + n[n.index("<"):n.index(">") + 1]
    if "irn::" in n:
        return "irn: " + n.split("irn::")[1].split("(anonymous namespace)::")[-1][:60]
    if "transpose" in n.lower():
        return "layout transposes"
    return "torch/other: " + n.replace("void at::native::", "").replace("(anonymous namespace)::", "")[:110]


rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
t, c = collections.Counter(), collections.Counter()
for r in rows:
    k = kind(r["Name"])
    t[k] += int(r["TotalDurationNs"])
    c[k] += int(r["Calls"])
print("total %.1f ms of kernel time" % (tot / 1e6))
for k, v in t.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    print("%6.2f%% %9.2f ms %6d  %s" % (100 * v / tot, v / 1e6, c[k], k))
