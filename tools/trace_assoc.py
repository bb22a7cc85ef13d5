#!/usr/bin/env python3
"""Per-launch durations of the association kernel from a rocprofv3 --kernel-trace CSV: grouped by position in the
ICP loop (launch 0 = first association of a localisation, ...).  usage: trace_assoc.py kernel_trace.csv [launches_per_loop]"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 11
k = [r for r in rows if "knn1_loop_k" in r["Kernel_Name"]]
k.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in k])
gap = np.array([(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(k[:-1], k[1:])])
n = len(d) // per * per
print("launches", len(d), "mean us %.2f" % d.mean(), "grid variant:", sum("ILb1" in r["Kernel_Name"] or "<true" in r["Kernel_Name"] for r in k))
m = d[:n].reshape(-1, per)
print("by position in the loop (mean us over %d loops):" % m.shape[0], " ".join("%.1f" % x for x in m.mean(0)))
print("last loop:", " ".join("%.1f" % x for x in m[-1]))
print("first loop:", " ".join("%.1f" % x for x in m[0]))
g = gap[:n - 1]
print("gap to the next association launch us: p50 %.2f mean %.2f" % (np.percentile(g[g < 50], 50), g[g < 50].mean()))
tot = m.sum(1)
nb = 10
print("per-loop association time (us), mean over consecutive tenths of the run:", " ".join("%.0f" % tot[i * len(tot) // nb:(i + 1) * len(tot) // nb].mean() for i in range(nb)))
