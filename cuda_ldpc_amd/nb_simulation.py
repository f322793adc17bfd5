"""Simulation and Eb/N0 sweep of the reference's non-binary program, driven through the C ABI.

Mirrors myNBLDPC/src/Simulation.cpp:16-87 (decode_once_cpu: channel -> Demodulate -> Decoding_EMS -> Statistic until
`num_Error_Frames >= leastErrorFrames && num_Frames >= leastTestFrames`) and src/main.cu:215-260 (sweep: seeds reset
to 173/173/173 at every point, sigma from Eb/N0 and the code rate).  The reference decodes one frame per call and tests
its stop rule after every frame; here frames are decoded in batches, the per-frame results are then accounted in stream
order and the accounting stops at the frame at which the reference would have stopped, so the counters are the ones a
single-threaded reference run (THREAD_NUM 1) produces.
"""
import numpy as np
import torch

from . import nbldpc as nb


class _SeedsAfter:
    """seeds_before[b] of a device-generated batch, computed on demand by jump-ahead."""

    def __init__(self, seed0, code, qam=False):
        self.seed0, self.code, self.qam = seed0, code, qam

    def __getitem__(self, b):
        return nb.seed_after(self.seed0, b, self.code, self.qam)


class NBSim:
    """The counters of class Simulation (include/struct.h:52-71) that the NB program uses."""

    def __init__(self, SNR=0.0):
        self.SNR = SNR
        self.num_Frames = 0
        self.num_Error_Frames = 0
        self.num_Error_Bits = 0   # symbol errors (the reference's name, Simulation.cpp:276)
        self.Total_Iteration = 0

    def row(self, N):
        n = max(self.num_Frames, 1)
        # Simulation.cpp:193-198: FER, "BER" = symbol errors / frames / N, AverageIT
        return " %.1f %8d  %4d  %6.4e  %6.4e  %.2f" % (self.SNR, self.num_Frames, self.num_Error_Frames, self.num_Error_Frames / n,
                                                  self.num_Error_Bits / n / N, self.Total_Iteration / n)


def Simulation_GPU(code, seed, sigma, SIM, CodeWord_sym, EMS_Nm=2, EMS_Nc=2, maxIT=20, batch=1024, leastErrorFrames=50,
                   leastTestFrames=1000, max_frames=None, device=None, device_channel=False, decoder_method=0, CONSTELLATION=None):
    """One Eb/N0 point. `seed` (int32[3]) advances exactly as far as the reference would have drawn.
    device_channel: generate the noise on the GPU (same uniforms, device libm) instead of the host, frame by frame.
    decoder_method (define.h:37, Simulation.cpp:54-70): 0 EMS, 1 trellis min-max, 2 log-QSPA = EMS(q, dc-1), 3 layered TMM.
    CONSTELLATION (host float32 [q, 2], Get_CONSTELLATION): the n_QAM = q branches of Modulate / AWGNChannel_CPU / Demodulate
    (one constellation point per code symbol) instead of BPSK; the caller passes the sigma of that n_QAM."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    cw = np.ascontiguousarray(CodeWord_sym, np.int32)
    cw_dev = torch.from_numpy(cw).to(device)
    qam = CONSTELLATION is not None
    con = np.ascontiguousarray(CONSTELLATION, np.float32) if qam else None
    con_dev = torch.from_numpy(con).to(device) if qam else None
    while True:
        if device_channel:
            seed0 = seed.copy()
            seeds_before = _SeedsAfter(seed0, code, qam)
            rxt = nb.AWGNChannel_GPU(seed, sigma, code, cw_dev, batch, CONSTELLATION=con_dev)
        else:
            seeds_before = []
            rx = np.empty((batch, code.N, 2) if qam else (batch, code.N * code.m), np.float32)
            for b in range(batch):
                seeds_before.append(seed.copy())
                rx[b] = nb.AWGNChannel_CPU(seed, sigma, code, cw, CONSTELLATION=con)
            rxt = torch.from_numpy(rx).to(device)
        Lch = nb.Demodulate(code, rxt, sigma, CONSTELLATION=con_dev)
        if decoder_method == 0:
            r = nb.Decoding_EMS(code, Lch, EMS_Nm, EMS_Nc, maxIT)
        elif decoder_method == 2:
            r = nb.Decoding_EMS(code, Lch, code.q, code.dc - 1, maxIT)  # Simulation.cpp:63-66
        else:
            r = nb.Decoding_TMM(code, Lch, maxIT, layered=(decoder_method == 3))
        errs = (r["DecodeOutput"] != cw_dev[None, :]).sum(dim=1).cpu().numpy()
        its = r["iter_number"].cpu().numpy()
        for b in range(batch):  # account in stream order; stop where the reference's while-condition fails
            SIM.num_Frames += 1
            SIM.Total_Iteration += int(its[b])
            SIM.num_Error_Bits += int(errs[b])
            SIM.num_Error_Frames += 1 if errs[b] else 0
            done = SIM.num_Error_Frames >= leastErrorFrames and SIM.num_Frames >= leastTestFrames
            if done or (max_frames is not None and SIM.num_Frames >= max_frames):
                if b + 1 < batch:
                    seed[:] = seeds_before[b + 1]  # the reference never drew the remaining frames
                return 1 if done else 0


def sweep(code, CodeWord_sym, startSNR=0.0, stopSNR=5.0, stepSNR=0.5, snrtype=0, seeds=(173, 173, 173), log=print, n_QAM=2, **kw):
    """main.cu:215-260: returns the list of NBSim, one per Eb/N0 point.  n_QAM enters sigma (main.cu:223); pass the
    constellation itself as CONSTELLATION=... for n_QAM != 2."""
    out = []
    s = np.float32(startSNR)
    while s <= stopSNR:
        seed = np.array(seeds, np.int32)
        SIM = NBSim(float(s))
        Simulation_GPU(code, seed, nb.sigma_of(float(s), code.rate, snrtype, n_QAM), SIM, CodeWord_sym, **kw)
        if log:
            log(SIM.row(code.N))
        out.append(SIM)
        s = np.float32(np.float64(s) + stepSNR)
    return out
