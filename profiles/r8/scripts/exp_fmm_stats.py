import os, sys, time, json, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_goal_gpu import _maze
from peanut_amd import _lib
from peanut_amd.goal import GeodesicSolver
lib = _lib.load()
raw = C.CDLL(lib._name)
for shape, seed in (((960, 960), 1),):
    trav = _maze(shape[0], shape[1], seed)
    sol = GeodesicSolver(shape[0], shape[1], 0)
    t = torch.from_numpy(trav).cuda()
    goal = tuple(int(v) for v in np.argwhere(trav)[len(np.argwhere(trav)) // 3])
    for rep in range(2):
        out = (C.c_ulonglong * 16)()
        raw.peanut_goal_debug_stats(None, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sol.distance(t, goal=goal)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        raw.peanut_goal_debug_stats(out, 0)
        o = list(out); n = max(o[0], 1)
        print(json.dumps({"ms": round(ms, 3), "rounds": sol.rounds, "tile_rounds": o[0], "avg": {"total": o[1] // n, "load": o[2] // n, "phase1": o[3] // n, "phase2": o[4] // n, "outer": round(o[5] / n, 2), "busiest_wave_iters": round(o[6] / n, 1), "busiest_wave_inner_cycles": o[7] // n},
                          "worst": {"total": o[8], "load": o[9], "phase1": o[10], "phase2": o[11], "outer": o[12], "busiest_wave_iters": o[13], "busiest_wave_inner_cycles": o[14]}}))
