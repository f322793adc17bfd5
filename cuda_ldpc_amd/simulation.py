"""Simulation_GPU and the Es/N0 sweep of the reference's binary program, driven through the C ABI.

Mirrors bldpc_实习/Simulation.cu:12-171 (batch loop: channel -> decode -> Statistic until the stop rule) and
main.cu:114-160 (sweep: seeds reset to 173/173/173 and counters cleared at every SNR point; SIM->SNR is a
float advanced by a double step).  Differences: shapes are arguments instead of define.cuh macros, the
statistics run on the device, and the batch may be sharded over ranks (sharding.py) with one all-reduce
of the counters per batch.
"""
import numpy as np
import torch

from . import sharding
from .bldpc import (EXIT_BATCH_GLOBAL, KERNEL_AUTO, AWGNChannel_CPU, AWGNChannel_GPU, Decode_Statistic, SimCounters, sigma_of)


def Simulation_GPU(code, seed, sigma, SIM, Num_Frames_OneTime=4096, maxIT=50, exit_mode=EXIT_BATCH_GLOBAL, kernel=KERNEL_AUTO,
                   leastErrorFrames=50, leastTestFrames=10000, displayStep=40960, dist=None, device=None, max_batches=None,
                   log=print, device_channel=False):
    """One SNR point (Simulation.cu:12-171).  `seed` (int32[3]) is the AWGN->seed state, advanced in place by the
    WHOLE batch on every rank so that all ranks stay on the reference's single noise stream."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    device = device or torch.device("cuda", torch.cuda.current_device())
    F = Num_Frames_OneTime
    first, count = sharding.shard_frames(F, world, rank)
    per_frame = sharding.binary_draws_per_frame(code.N)
    dev_cnt = torch.zeros(5, dtype=torch.int64, device=device)
    D = torch.empty((code.N + 1, max(count, 1)), dtype=torch.int32, device=device)
    length = code.K  # Message_CW 0 (define.cuh:61)
    batches = 0
    while True:
        SIM.num_Frames += F  # Simulation.cu:113
        my_seed = sharding.lcg_jump(seed, first * per_frame)
        if not count:
            yd = None
        elif device_channel:  # same draws, generated on the GPU (device libm in the Box-Muller transform)
            yd = AWGNChannel_GPU(my_seed, sigma, code.N, count, device=device)
        else:
            yd = torch.from_numpy(AWGNChannel_CPU(my_seed, sigma, code.N, count)).to(device)
        seed[:] = sharding.lcg_jump(seed, F * per_frame)
        dev_cnt.zero_()
        if count:
            Decode_Statistic(code, yd, dev_cnt, max_iter=maxIT, length=length, exit_mode=exit_mode, kernel=kernel, D=D)  # Simulation.cu:143-145
        sharding.allreduce_counters(dev_cnt, dist)
        c = dev_cnt.cpu().tolist()
        SIM.num_Error_Frames += c[0]
        SIM.num_Error_Bits += c[1]
        SIM.Total_Iteration += c[2]
        SIM.num_False_Frames += c[3]
        SIM.num_Alarm_Frames += c[4]
        batches += 1
        stop = SIM.num_Error_Frames >= leastErrorFrames and SIM.num_Frames >= leastTestFrames
        last = max_batches is not None and batches >= max_batches
        if rank == 0 and log and (SIM.num_Frames % displayStep == 0 or stop or last):
            log(format_row(SIM, length))
        if stop or (max_batches is not None and batches >= max_batches):
            return 1 if stop else 0


def format_row(SIM, length):
    """The reference's result row (Simulation.cu:272): SNR NTF NEF FER BER AverIT FER_F FER_A."""
    r = SIM.ratios(length)
    return " %.1f %8d  %4d  %6.4e  %6.4e  %.2f  %6.4e %6.4e" % (SIM.SNR, SIM.num_Frames, SIM.num_Error_Frames, r["FER"], r["BER"],
                                                           r["AverageIT"], r["FER_False"], r["FER_Alarm"])


def snr_grid(startSNR=0.0, stopSNR=13.0, stepSNR=0.2):
    """SNR points the reference visits: a float32 accumulated with a double step (main.cu:114)."""
    pts, s = [], np.float32(startSNR)
    while s <= stopSNR:
        pts.append(float(s))
        s = np.float32(np.float64(s) + stepSNR)
    return pts


def sweep(code, startSNR=0.0, stopSNR=13.0, stepSNR=0.2, snrtype=1, seeds=(173, 173, 173), dist=None, log=print, **kw):
    """main.cu:114-160: returns the list of SimCounters, one per SNR point."""
    out = []
    for snr in snr_grid(startSNR, stopSNR, stepSNR):
        seed = np.array(seeds, np.int32)  # reset at every point (main.cu:117-119)
        SIM = SimCounters()
        SIM.SNR = snr
        Simulation_GPU(code, seed, sigma_of(snr, snrtype, code.K / code.N), SIM, dist=dist, log=log, **kw)
        out.append(SIM)
    return out
