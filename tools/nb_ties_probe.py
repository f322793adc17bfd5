import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuda_ldpc_amd import nbldpc as nb
from oracle import pyoracle as orc
nbd = os.path.join(ROOT, "data", "nb")
mul, _, _ = nb.GFInitial(64, os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
code = nb.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), mul)
ocode = orc.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
rng = np.random.default_rng(3)
B = 4
Lch = rng.integers(-3, 4, size=(B, code.N, code.q - 1)).astype(np.float32)
Lch[0, :, ::5] = -0.0
Lch[1] *= 1e30
Lch[2] *= 1e-40
for maxit in (1, 2, 5):
    r = nb.Decoding_EMS(code, torch.from_numpy(Lch).cuda(), 2, 2, maxit, want_state=True)
    torch.cuda.synchronize()
    for b in range(B):
        want = orc.nb_ems_decode(ocode, Lch[b], 2, 2, maxit, want_state=True)
        out = r["DecodeOutput"][b].cpu().numpy()
        llr = r["LLR"][b].cpu().numpy()
        c2v = r["L_c2v"][b].cpu().numpy()
        print("maxit", maxit, "b", b, "out diff", int((out != want["out"]).sum()), "LLR diff", int((llr.view(np.uint32) != want["LLR"].view(np.uint32)).sum()),
              "c2v diff", int((c2v.view(np.uint32) != want["c2v"].reshape(c2v.shape).view(np.uint32)).sum()) if "c2v" in want else "-", "it", int(r["iter_number"][b]), want["it"])
