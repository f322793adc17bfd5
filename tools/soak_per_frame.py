"""Soak of the early-exit paths at full batch size: (1) repeated per-frame decodes of the same batch give identical
results (a race in the flag words or in the retire path would not), (2) fused kernels and table kernels agree frame by
frame in per-frame mode, (3) batch-global agrees between the two kernel families, (4) so do 50 fixed iterations, hard bits and a-posteriori sums.   usage: python tools/soak_per_frame.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_ldpc_amd as C  # noqa: E402

CASES = [("J4_L24_Z96_BlockH.txt", 4, 24, 96, 16384, (2.4, 3.0, 3.8)), ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 8192, (-1.0, -0.4)),
         ("J10_L60_Z160_BlockH.txt", 10, 60, 160, 2048, (2.6,)), ("PON_LDPC.txt", 12, 69, 256, 1024, (2.2,)),
         ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, 512, (-1.4,)), ("J4_L24_Z512_BlockH.txt", 4, 24, 512, 2048, (3.2,))]
bad = 0
for fn, J, L, Z, F, snrs in CASES:
    code = C.BinaryCode.from_blockh(os.path.join(ROOT, "data", "bldpc", fn), J, L, Z)
    for snr in snrs:
        seed = np.array([173, 173, 173], np.int32)
        y = C.AWGNChannel_GPU(seed, C.sigma_of(snr), code.N, F)
        ref = None
        for rep in range(6):
            r = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_PER_FRAME, kernel=C.KERNEL_QC_LDS)
            torch.cuda.synchronize()
            cur = (r["D"].clone(), r["iters"].clone())
            if ref is None:
                ref = cur
            elif not (torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])):
                bad += 1
                print("NOT REPRODUCIBLE", fn, snr, rep)
        kname = code.last_kernel
        t = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_PER_FRAME, kernel=C.KERNEL_TABLE)
        torch.cuda.synchronize()
        same = torch.equal(t["D"], ref[0]) and torch.equal(t["iters"], ref[1])
        g1 = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL, kernel=C.KERNEL_QC_LDS)
        g2 = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL, kernel=C.KERNEL_TABLE)
        torch.cuda.synchronize()
        same_g = g1["iteraTime"] == g2["iteraTime"] and torch.equal(g1["D"], g2["D"])
        # (4) fixed iterations at full batch size: the kernels bench.py times (local edges where the code has them), a-posteriori sums included
        f1 = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_FIXED, kernel=C.KERNEL_QC_LDS, want_app=True)
        kfixed = code.last_kernel
        f2 = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_FIXED, kernel=C.KERNEL_TABLE, want_app=True)
        torch.cuda.synchronize()
        same_f = torch.equal(f1["D"], f2["D"]) and torch.equal(f1["app"].view(torch.int32), f2["app"].view(torch.int32))
        bad += (not same) + (not same_g) + (not same_f)
        it = ref[1].float()
        print("%-26s Es/N0 %5.1f F=%d: per-frame fused==table %s (mean %.2f it, max %d), batch-global fused==table %s (stops at %d) [%s]; fixed 50 fused==table (bits and sums) %s [%s]"
              % (fn, snr, F, same, it.mean().item(), int(ref[1].max()), same_g, g1["iteraTime"], kname, same_f, kfixed), flush=True)
print("soak:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
