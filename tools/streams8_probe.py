"""configs[4] as eight concurrent streams on one GPU: the variant each stream's launch runs.
PAGK_QUAD_MIN out of reach = the 4-wave kernel; default = four features per wave (8 x 4000 features with the
concurrency hint); PAGK_LEVELS_SHARED=1 = four features per wave, one level per wave."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
for name, env in (("4-wave kernel", {"PAGK_QUAD_MIN": "1000000000"}), ("four features per wave", {}),
                  ("one level per wave", {"PAGK_LEVELS_SHARED": "1"}), ("four features per wave", {}),
                  ("one level per wave", {"PAGK_LEVELS_SHARED": "1"})):
    for k in ("PAGK_QUAD_MIN", "PAGK_LEVELS_SHARED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    r = bench.config_row(4, 4000, 0, 30, streams=8)
    print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "workload"}, flush=True)
