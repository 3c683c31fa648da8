# tools/kpp_sweep.py: every case of tests/test_gpu_kpp_pick.py against the numpy reference, printing the cases that differ (a debugging aid:
# the test stops at the first)
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pq_vector_amd as pqv
import test_gpu_kpp_pick as T
rng = np.random.default_rng(6)
nbad = 0; ntot = 0
for name, md in T._cases(rng):
    for workers in (1, 3, 8, 64, 1000):
        for draw in (0.0, 0.37, 0.99999994):
            total, pick = T._reference(md, workers, draw)
            d_total, d_pick, status = T._device(pqv, md, workers, draw)
            ntot += 1
            ok = True
            if not (total > 0 and np.isfinite(total)): ok = status == 1
            elif d_total.view(np.uint32) != total.view(np.uint32): ok = False
            elif pick is None: ok = status == 3
            else: ok = status == 0 and d_pick == pick
            if not ok:
                nbad += 1
                if nbad < 60: print("BAD", name, len(md), workers, draw, "ref", total, pick, "dev", d_total, d_pick, status)
print("cases", ntot, "bad", nbad)
