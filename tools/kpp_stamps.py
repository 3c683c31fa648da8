# tools/kpp_stamps.py: PQV_KPP_STAMPS=1 python tools/kpp_stamps.py -- the 100 MHz clock at a few places of one kpp_pick_kernel launch
# (pqv_kpp_pick prints them to stderr: chain block, last summary block, chunk block 0, the end of every wave turn) on 50 000 minima, 256 / 8 worker chunks
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pq_vector_amd as pqv
import test_gpu_kpp_pick as T
rng = np.random.default_rng(1)
md = (rng.random(50000, dtype=np.float32) * 100).astype(np.float32)
for workers in (256, 8):
    for rep in range(3):
        print("workers", workers, T._device(pqv, md, workers, 0.7), flush=True)
