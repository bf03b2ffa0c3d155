#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_r1.sh) into profiles/<round>/:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
   <tag>_pmc.csv            per-launch FETCH_SIZE / WRITE_SIZE of the RX kernels (separate --pmc passes)
   traffic.json             hbm bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (calibration: README.md)
"""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
out = os.path.join(root, "profiles", rnd)
os.makedirs(out, exist_ok=True)
traffic = {}
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_*"))):
    tag = os.path.basename(d)[5:]
    st = os.path.join(d, "trace", "t_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(out, tag + "_kernel_stats.csv"))
    rows = []
    per = {}
    for name, sub, pre in (("FETCH_SIZE", "pmc_fetch", "f"), ("WRITE_SIZE", "pmc_write", "w")):
        f = os.path.join(d, sub, pre + "_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if r["Counter_Name"] == name and ("k_ssb" in k or "k_cw" in k or "k_hilb" in k or "generic" in k or "k_tx" in k):
                rows.append((k.split("(")[0], name, r["Dispatch_Id"], r["Counter_Value"], r["VGPR_Count"], r["LDS_Block_Size"]))
                per.setdefault((k.split("(")[0], name), []).append(float(r["Counter_Value"]))
    with open(os.path.join(out, tag + "_pmc.csv"), "w") as fo:
        fo.write("kernel,counter,dispatch,value_KiB,vgpr,lds_bytes\n")
        for r in rows:
            fo.write(",".join('"%s"' % x if i == 0 else str(x) for i, x in enumerate(r)) + "\n")
    # (SELENITE_ARITH_AUTO launches two kernels per call, the second one empty in the steady state: keep the one that moves the bytes)
    kernels = sorted({k for k, _ in per}, key=lambda k: sum(per.get((k, "FETCH_SIZE"), [0])) / max(len(per.get((k, "FETCH_SIZE"), [0])), 1))
    for k in kernels:
        fs = per.get((k, "FETCH_SIZE")); ws = per.get((k, "WRITE_SIZE"))
        if fs and ws:
            rd, wr = 2 * 1024 * sum(fs) / len(fs), 1024 * sum(ws) / len(ws)
            ms = None
            if os.path.exists(st):
                for r in csv.DictReader(open(st)):
                    if r["Name"].split("(")[0] == k:
                        ms = float(r["AverageNs"]) / 1e6
            traffic[tag] = {"kernel": k, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                            "rocprof_avg_ms": ms}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for t, v in traffic.items():
    print("%-14s %-46s rd %.3f GB wr %.3f GB avg %.4f ms" % (t, v["kernel"][-46:], v["hbm_read_bytes"] / 1e9, v["hbm_write_bytes"] / 1e9, v["rocprof_avg_ms"] or -1))
