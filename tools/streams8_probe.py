import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
for qmin in ("1000000000", "4000"):
    os.environ["PAGK_QUAD_MIN"] = qmin
    r = bench.config_row(4, 4000, 0, 30, streams=8)
    print("PAGK_QUAD_MIN", qmin, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "workload"}, flush=True)
