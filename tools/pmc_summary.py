#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel and counter, mean value per dispatch."""
import glob
import sys

import pandas as pd

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
rows = []
for f in sorted(glob.glob(root + "/*/*counter_collection.csv")):
    df = pd.read_csv(f)
    g = df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()
    rows.append(g)
out = pd.concat(rows)
out = out[out.Kernel_Name.str.contains("dort")]
pd.set_option("display.width", 200)
print("# counters per dispatch of the pair kernel (mean over dispatches); source:", root)
for _, r in out.iterrows():
    print("%-28s %20.1f   (n=%d)" % (r.Counter_Name, r["mean"], r["count"]))
