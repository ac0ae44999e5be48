#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh: per kernel of the pipeline, every counter (mean per
dispatch) and the derived figures the design discussion uses -- HBM traffic per pipeline launch (2 x FETCH_SIZE +
WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reports half of the bytes of wide coalesced reads),
matrix-core busy fraction, share of wave cycles spent waiting, VALU instructions per launch.

    python tools/pmc_summary.py gpurun_out/pmc [round-tag [config solves_per_launch]]
        prints the summary; with a tag also writes the entry of that bench configuration (default 1: the headline, 5120
        solves per launch) into profiles/hbm_traffic.json
"""
import glob
import json
import os
import sys

import pandas as pd

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
tag = sys.argv[2] if len(sys.argv) > 2 else None
config = sys.argv[3] if len(sys.argv) > 3 else "1"
solves = int(sys.argv[4]) if len(sys.argv) > 4 else 5120
frames = []
for f in sorted(glob.glob(root + "/*/*counter_collection.csv")):
    df = pd.read_csv(f)
    df["pass"] = os.path.basename(os.path.dirname(f))
    frames.append(df)
df = pd.concat(frames)
df = df[df.Kernel_Name.str.contains("dort")]
df["kernel"] = df.Kernel_Name.str.extract(r"(dort_[a-z0-9_]+)")[0]
# mean per dispatch of every instantiation, then the SUM over the instantiations of a kernel that run in one pipeline launch
# (the Jacobi kernel is launched once per size class of items: k_jacobi.hip)
tab = (df.groupby(["kernel", "Kernel_Name", "Counter_Name"])["Counter_Value"].mean()
         .groupby(level=["kernel", "Counter_Name"]).sum().unstack(0))
pd.set_option("display.width", 250)
pd.set_option("display.float_format", lambda v: "%.4g" % v)
print("# rocprofv3 --pmc, per pipeline launch: mean per dispatch of each kernel, summed over the size-class launches of the Jacobi kernel (bench.py --config %s, %d solves per launch); source:" % (config, solves), root)
print(tab.to_string())
print()
print("# derived, per kernel")
out = {}
for k in tab.columns:
    c = tab[k]
    line = ["%-22s" % k]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # KB
        hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        out[k] = hbm
        line.append("HBM bytes %.3e (2 x FETCH + WRITE)" % hbm)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        line.append("waiting %.0f %% of wave cycles, issuing %.0f %%" % (100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                                                                        100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"] > 0:
        # MFMA_BUSY counts cycles per SIMD-pipe, BUSY_CU_CYCLES per CU: four matrix pipes per CU
        line.append("matrix core busy %.1f %% of the CU-busy cycles x 4 pipes" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_BUSY_CU_CYCLES"])))
    if "SQ_INSTS_VALU_MFMA_F64" in c:
        line.append("FP64 MFMA %.3e (%.3e flop)" % (c["SQ_INSTS_VALU_MFMA_F64"], 2048 * c["SQ_INSTS_VALU_MFMA_F64"]))
    if "SQ_INSTS_VALU" in c:
        line.append("VALU insts %.3e" % c["SQ_INSTS_VALU"])
    print("  ".join(line))
if out and tag:
    total = float(sum(out.values()))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")
    fetch = float(sum(tab[k]["FETCH_SIZE"] for k in out) * 1024)
    write = float(sum(tab[k]["WRITE_SIZE"] for k in out) * 1024)
    entry = {"round": tag, "solves_per_launch": solves, "traffic_bytes_per_launch": total, "fetch_size_bytes_raw": fetch,
             "write_size_bytes": write, "per_kernel_bytes": {k: float(v) for k, v in out.items()},
             "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB) summed over the pipeline kernels of one launch; traffic = "
                     "2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); collected with "
                     "tools/pmc_passes.sh" + (", summary in profiles/%s_pmc_counters.txt" % tag if config == "1" else "")}
    doc = {}
    if os.path.exists(path):
        doc = json.load(open(path))
    if config == "1":      # the headline entry stays at the top level (what bench.py has read since round 2)
        doc = {**entry, "configs": doc.get("configs", {})}
    else:
        doc.setdefault("configs", {})[config] = entry
    json.dump(doc, open(path, "w"), indent=1)
    print("\n# wrote", os.path.relpath(path), "traffic per launch %.4e bytes" % total)
