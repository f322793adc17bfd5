"""Rate of the device-side channel generators.   usage: python tools/channel_time.py   (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_ldpc_amd as C  # noqa: E402

for N, F in ((2304, 65536), (4096, 65536), (38400, 8192)):
    seed = np.array([173, 173, 173], np.int32)
    y = C.AWGNChannel_GPU(seed, 0.5, N, F)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = C.AWGNChannel_GPU(seed, 0.5, N, F)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("bldpc_awgn_channel_device N=%d F=%d: %.3f ms per batch, %.1f G samples/s, %.1f M frames/s" % (N, F, dt * 1e3, N * F / dt / 1e9, F / dt / 1e6), flush=True)
