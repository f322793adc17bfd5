"""Soak of the local-edge kernels' host side (matching, block / row ordering, sorted place codes) on RANDOM block patterns:
J4_L24_Z96-shaped matrices with full rows and J32_L64_Z64-shaped matrices with columns of weight 3 (row weights 5 ... 7), random
shifts; the fused kernel (local edges where the matching exists) against the table kernel, hard bits and a-posteriori sums, 1 / 2 / 9
fixed iterations, and the batch-global rule.   usage: python tools/local_edge_soak.py [matrices per shape]   (GPU box)"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_ldpc_amd as C  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2025)
bad = local = total = 0
tmp = tempfile.mkdtemp()
for shape in ("J4", "J32"):
    for m in range(n):
        if shape == "J4":
            J, L, Z, snr = 4, 24, 96, 3.0
            while True:
                H = rng.integers(0, Z, size=(J, L))
                for j in range(J):
                    H[j, rng.permutation(L)[:4]] = -1
                if ((H >= 0).sum(0) >= 1).all():
                    break
        else:
            J, L, Z, snr = 32, 64, 64, 0.5
            while True:  # three layers, each a random 2-to-1 map of the columns onto the rows: column weight 3, row weight 6
                H = -np.ones((J, L), np.int64)
                ok = True
                for _ in range(3):
                    rows = np.repeat(np.arange(J), 2)[rng.permutation(L)]
                    for l in range(L):
                        if H[rows[l], l] >= 0:
                            ok = False
                        H[rows[l], l] = rng.integers(0, Z)
                if not ok:
                    continue
                if m % 2:  # move a few blocks inside their columns: rows of 5 and 7
                    for _ in range(3):
                        l = int(rng.integers(0, L))
                        src = [j for j in range(J) if H[j, l] >= 0 and (H[j] >= 0).sum() == 6]
                        dst = [j for j in range(J) if H[j, l] < 0 and (H[j] >= 0).sum() == 6]
                        if src and dst:
                            a, b = src[0], dst[int(rng.integers(0, len(dst)))]
                            H[b, l] = H[a, l]
                            H[a, l] = -1
                if ((H >= 0).sum(0) == 3).all() and (H >= 0).sum(1).max() <= 7 and (H >= 0).sum(1).min() >= 5:
                    break
        path = os.path.join(tmp, "H.txt")
        with open(path, "w") as f:
            for row in H:
                f.write("\t".join(str(int(x)) for x in row) + "\r\n")
        code = C.BinaryCode.from_blockh(path, J, L, Z)
        F = 6 if m % 3 else 5
        seed = np.array([173 + m, 173, 173], np.int32)
        y = C.AWGNChannel_GPU(seed, C.sigma_of(snr), code.N, F)
        same = True
        for its in (1, 2, 9):
            a = C.LDPC_Decoder_GPU(code, y, max_iter=its, exit_mode=C.EXIT_FIXED, kernel=C.KERNEL_QC_LDS, want_app=True)
            name = code.last_kernel
            b = C.LDPC_Decoder_GPU(code, y, max_iter=its, exit_mode=C.EXIT_FIXED, kernel=C.KERNEL_TABLE, want_app=True)
            torch.cuda.synchronize()
            same = same and torch.equal(a["D"], b["D"]) and torch.equal(a["app"].view(torch.int32), b["app"].view(torch.int32))
        g1 = C.LDPC_Decoder_GPU(code, y, max_iter=40, exit_mode=C.EXIT_BATCH_GLOBAL, kernel=C.KERNEL_QC_LDS, want_app=True)
        g2 = C.LDPC_Decoder_GPU(code, y, max_iter=40, exit_mode=C.EXIT_BATCH_GLOBAL, kernel=C.KERNEL_TABLE, want_app=True)
        torch.cuda.synchronize()
        same = same and g1["iteraTime"] == g2["iteraTime"] and torch.equal(g1["D"], g2["D"]) and torch.equal(g1["app"].view(torch.int32), g2["app"].view(torch.int32))
        total += 1
        local += "local" in name
        bad += not same
        w = (H >= 0).sum(1)
        print("%s #%d rows %d..%d cols %d..%d F=%d: fused==table %s [%s]" % (shape, m, w.min(), w.max(), (H >= 0).sum(0).min(), (H >= 0).sum(0).max(), F, same, name.split("<")[0]), flush=True)
print("local-edge soak: %d matrices, %d on the local-edge kernels, %s" % (total, local, "OK" if bad == 0 else "%d MISMATCHES" % bad))
sys.exit(1 if bad else 0)
