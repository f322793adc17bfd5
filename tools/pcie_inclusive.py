"""Experiment: end-to-end rate of BASELINE config 2 when the boundary hands over HOST buffers (pinned):
H2D of Channel_Out, bldpc_decode, D2H of D (the reference's LDPC_Decoder_GPU contract: device input, host D)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_ldpc_amd as C
J, L, Z, F = 4, 24, 96, 65536
code = C.BinaryCode.from_blockh(os.path.join(ROOT, "data", "bldpc", "J4_L24_Z96_BlockH.txt"), J, L, Z)
N = L * Z
yh = (torch.randn((N, F)) * 0.5 + 1.0).pin_memory()
Dh = torch.empty((N + 1, F), dtype=torch.int32).pin_memory()
yd = torch.empty((N, F), device="cuda")
Dd = torch.empty((N + 1, F), dtype=torch.int32, device="cuda")
def step(h2d, d2h):
    if h2d:
        yd.copy_(yh, non_blocking=True)
    C.LDPC_Decoder_GPU(code, yd, max_iter=50, exit_mode=C.EXIT_FIXED, D=Dd)
    if d2h:
        Dh.copy_(Dd, non_blocking=True)
for name, h2d, d2h in (("device in, device out", 0, 0), ("device in, host D (the reference contract)", 0, 1), ("host in, host D", 1, 1)):
    step(h2d, d2h); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step(h2d, d2h)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-45s %.2f ms per 65536 codewords = %.2f M codewords/s" % (name, dt * 1e3, F / dt / 1e6))
