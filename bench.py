#!/usr/bin/env python3
"""bench.py -- decoded codewords/s of the LDPC hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--kernel auto|qc|table]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the decode hot path (bldpc_decode: 50 flooding min-sum iterations, early exit
off, followed by the device-side error statistics) over one batch of synthetic channel values that are
already resident in HBM.  Workload (BASELINE.json configs[1]): J4_L24_Z96 rate-5/6, 65536 codewords
per GPU, Es/N0 = 3.0 dB, all-zero codeword, reference AWGN stream (seeds 173/173/173): a 4096-frame
block generated once on the host and tiled to the batch.  Frames shard across ranks
with no data-path collective (weak scaling); RCCL carries only the final all-reduce of the five error
counters.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against HBM (the bound
SURVEY.md 8d prescribes): algorithmic bytes = (8N+4) per codeword * codewords per launch over the
kernel's average duration measured with HIP events around the decode call alone.  `cpu_baseline` times
the CPU oracle (a port of the reference kernels -- the binary reference has no CPU decode path) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    # name: (file, J, L, Z, frames per GPU, Es/N0 dB, iterations)
    "J4_L24_Z96": ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 65536, 3.0, 50),
    "J32_L64_Z64": ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 32768, 0.0, 50),
    "J15_L30_Z1280": ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, 1024, 0.0, 50),
    # further matrices of the reference (not BASELINE configs): its compiled-in default (define.cuh:20-22) and one per family
    "PON_J12_L69_Z256": ("PON_LDPC.txt", 12, 69, 256, 8192, 2.0, 50),
    "J10_L60_Z160": ("J10_L60_Z160_BlockH.txt", 10, 60, 160, 16384, 2.5, 50),
    "J4_L24_Z512": ("J4_L24_Z512_BlockH.txt", 4, 24, 512, 8192, 3.0, 50),
    # GF(64) EMS (BASELINE.json configs[4]): frames per GPU, Eb/N0 dB, maxIT (reference default 20, define.h:35)
    "NB_BDS_GF64": ("BDS.576.288.GF.64.txt", 0, 0, 0, 16384, 3.0, 20),
    # the reference's other decoder_method values on the same code (define.h:37): 1 trellis min-max, 3 layered trellis min-max
    "NB_BDS_GF64_TMM": ("BDS.576.288.GF.64.txt", 0, 0, 1, 16384, 3.0, 20),
    "NB_BDS_GF64_LTMM": ("BDS.576.288.GF.64.txt", 0, 0, 3, 16384, 3.0, 20),
    # the reference's GF(256) code (12 symbols, 6 checks; second field = q): a message vector spans four waves (k_nb_ems_wide)
    "NB_N96_GF256": ("LDPC_N96_K48_GF256_d1_exp.txt", 256, 0, 0, 8192, 4.0, 20),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)   # 40 x 5.5 ms: the first ~6 launches after an idle period run 2-18 % slow
    ap.add_argument("--warmup", type=int, default=10)  # (clock ramp; profiles/r02c_binary_kernel_stats.csv shows the same tail)
    ap.add_argument("--workload", default="J4_L24_Z96", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "qc", "table"])
    ap.add_argument("--frames", type=int, default=0, help="override frames per GPU")
    ap.add_argument("--snr", type=float, default=None, help="override the SNR point (experiments only)")
    ap.add_argument("--iters", type=int, default=0, help="override the iteration count (experiments only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the cpu_baseline sample (0 = auto)")
    return ap.parse_args()


def cpu_baseline(name, J, L, Z, snr, iters, y_block, nframes):
    """Oracle (CPU port of the reference kernels, OpenMP over frames) on a bounded sample. Checker code,
    used here ONLY as the reported CPU baseline -- never on the product path."""
    from oracle import pyoracle as orc
    cores = cpu_threads()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    code = orc.BinaryCode(os.path.join(ROOT, "data", "bldpc", name), J, L, Z)
    if nframes <= 0:
        # ~115 codewords/s/core for J4_L24_Z96 (E = 7680 edges) at 50 iterations, inversely proportional to the edge
        # count and the iterations: size the sample for ~13 s of CPU work
        nframes = max(16, int(13.0 * 115.0 * cores * (7680.0 / (int((code.H != -1).sum()) * Z)) * (50.0 / max(iters, 1))) // 16 * 16)
    reps = (nframes + y_block.shape[1] - 1) // y_block.shape[1]
    y = np.ascontiguousarray(np.tile(y_block, (1, reps))[:, :nframes]).reshape(-1)
    orc.bldpc_decode(code, y, nframes, 1, early_exit=0)  # touch pages / spin up the OpenMP threads
    t0 = time.perf_counter()
    orc.bldpc_decode(code, y, nframes, iters, early_exit=0)
    dt = time.perf_counter() - t0
    return {"value": nframes / dt, "unit": "codewords/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same tiled 4096-frame block, %d iterations, oracle/bldpc_oracle.c (port: the binary reference has no CPU path), OpenMP over frames, %.1f s"
                      % (nframes, iters, dt)}


def onchip_ceiling(kernel_name, cycles_per_launch, frames, iters):
    """What actually bounds the fused kernels: the load of the VALU and of the LDS pipe per CU-iteration (one flooding
    iteration of the frames a CU holds at a time), from the committed micro-benchmarks (profiles/onchip_model.json names the
    rates and the instruction mix per kernel), against the cycles the kernel really takes: GRBM_GUI_ACTIVE / 8 per launch from
    the committed SQ pass, over (frames / frames-per-CU / 256 CUs) * iters CU-iterations -- prologue, epilogue and the tail of
    the launch included.  frac = the slower pipe's load / measured: 1.0 would be a kernel that does nothing but feed that pipe."""
    try:
        models = json.load(open(os.path.join(ROOT, "profiles", "onchip_model.json")))
    except Exception:
        return None
    for key, m in models.items():
        if not kernel_name.startswith(key) or not frames or not iters:
            continue
        cu_iters = frames / m["frames_per_cu"] / 256.0 * iters
        measured = cycles_per_launch / cu_iters
        pipe = max(m["valu_cycles_per_cu_iter"], m["lds_cycles_per_cu_iter"])
        out = {"valu_cycles_per_cu_iter": m["valu_cycles_per_cu_iter"], "lds_cycles_per_cu_iter": m["lds_cycles_per_cu_iter"],
               "measured_cycles_per_cu_iter": measured, "frac": pipe / measured, "model_source": m["source"]}
        if m.get("loop_cycles_per_cu_iter_stamps"):  # the iteration loop alone, from in-kernel time stamps (no prologue / tail)
            out["loop_cycles_per_cu_iter_stamps"] = m["loop_cycles_per_cu_iter_stamps"]
            out["frac_of_loop"] = pipe / m["loop_cycles_per_cu_iter_stamps"]
        return out
    return None


def pmc_onchip(kernel_name, kern_ms, frames=0, iters=0):
    """On-chip picture of the dominant kernel, looked up in the committed SQ counter passes (profiles/*_pmc.json "sq") --
    NOT measured in this run; "source" names the file.  The fused kernels are bound by VALU issue and the LDS pipe, not by
    HBM.  Reported: the wave-cycle split the guide defines (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, with
    WAIT_INST_LDS a sub-bucket of WAIT_INST_ANY), instruction counts per SIMD-cycle, and the pipe loads those counts imply at
    the issue rates measured in profiles/r02_micro_rates.txt (full-rate VALU 2.5 cycles, half-rate 4.2, ds_read_b64 2.1,
    ds_write_b64 6.2) -- SQ_ACTIVE_INST_VALU is one quad-cycle per instruction on gfx950 and says nothing about pipe time."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            j = json.load(open(p))
        except Exception:
            continue
        sq = j.get("sq") or {}
        if j.get("kernel") != kernel_name or "SQ_INSTS_VALU" not in sq:
            continue
        cycles = j["avg_ms"] * 1e-3 * 2.3e9 if not sq.get("GRBM_GUI_ACTIVE") else sq["GRBM_GUI_ACTIVE"] / 8.0
        wc = sq.get("SQ_WAVE_CYCLES")
        split = None
        if wc and sq.get("SQ_WAIT_ANY") is not None:
            split = {k: sq[c] / wc for k, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"),
                                                ("active_inst_any", "SQ_ACTIVE_INST_ANY"), ("wait_inst_lds", "SQ_WAIT_INST_LDS")) if sq.get(c) is not None}
        return {"wave_cycle_split": split, "valu_instructions_per_simd_cycle": sq["SQ_INSTS_VALU"] / 1024.0 / cycles,
                "lds_instructions_per_cu_cycle": (sq["SQ_INSTS_LDS"] / 256.0 / cycles) if sq.get("SQ_INSTS_LDS") else None,
                "lds_array_util": sq["SQ_LDS_IDX_ACTIVE"] / 256.0 / cycles, "lds_bank_conflict_cycles": sq.get("SQ_LDS_BANK_CONFLICT"),
                "ceiling": onchip_ceiling(kernel_name, cycles, frames, iters),
                "source": "committed profile " + os.path.basename(p),
                "note": "profiled pass of an earlier run of this command, not this run; kernel cycles from %s" % ("GRBM_GUI_ACTIVE/8" if sq.get("GRBM_GUI_ACTIVE") else "avg_ms * 2.3 GHz")}
    return None


def pmc_traffic(kernel_name):
    """(HBM bytes per launch of the dominant kernel, file it came from): a look-up into the committed rocprofv3 PMC passes
    (profiles/*_pmc.json, written by profiles/summarize.py from separate FETCH_SIZE / WRITE_SIZE runs of this same
    command), not a measurement of this run; (None, None) when no summary exists for this kernel."""
    import glob
    best = (None, None)
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            j = json.load(open(p))
        except Exception:
            continue
        if j.get("kernel") == kernel_name and j.get("hbm_bytes_per_launch"):
            best = (j["hbm_bytes_per_launch"], "committed profile " + os.path.basename(p))
    return best


def cpu_threads():
    """Host threads for the CPU baseline: this box's share for one GPU is 16 cores (gpurun contract)."""
    return max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("BENCH_CPU_THREADS", "16"))))


def nb_cpu_baseline(snr, nframes, method=0, q=64, matrix="BDS.576.288.GF.64.txt"):
    """The REFERENCE's own CPU decoder (oracle/_ref/nb_ref, built from /root/reference/myNBLDPC/src/*.cpp;
    single-threaded like the reference's THREAD_NUM 1) on the first frames of the same stream."""
    import subprocess
    from oracle import pyoracle as orc
    ref = orc.ref_binary()
    if q == 256:  # the reference built for GFQ 256 / LDPC_N96_K48_GF256_d1_exp.txt (oracle/Makefile), all-zero codeword
        ref = os.path.join(ROOT, "oracle", "_ref", "nb_ref_gf256")
        ref = ref if os.path.exists(ref) else None
    if ref is None:
        # the built reference binary did not travel: time the restatement (oracle/nbldpc_oracle.c) on the same frames, one core
        nbd = os.path.join(ROOT, "data", "nb")
        ocode = orc.NBCode(os.path.join(nbd, matrix), os.path.join(nbd, "GF", "Arith.Table.GF.%d.txt" % q))
        cw = np.loadtxt(os.path.join(nbd, "codeword_bds_gf64.txt"), dtype=np.int32) if q == 64 else np.zeros(ocode.N, np.int32)
        seed = np.array([173, 173, 173], np.int32)
        n = (nframes or 100) * (1 if method == 0 else 3)
        sigma = orc.nb_sigma(snr, ocode.rate)
        Lch = [orc.nb_channel(ocode, cw, seed, sigma)[1] for _ in range(n)]
        t0 = time.perf_counter()
        its = 0
        for L in Lch:
            r = orc.nb_ems_decode(ocode, L, 2, 2, 20) if method == 0 else orc.nb_tmm_decode(ocode, L, 20, layered=(method == 3))
            its += r["it"]
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "codewords/s", "cores": 1, "kind": "port",
                "sample": "first %d frames of the same seed-173 stream at Eb/N0 %.1f dB, oracle/nbldpc_oracle.c (the reference binary "
                          "oracle/_ref/nb_ref is not present), mean %.2f iterations, %.1f s" % (n, snr, its / n, dt)}
    nframes = nframes or 300  # ~33 frames/s at 3 dB -> ~10 s
    nframes = nframes * (1 if method == 0 else 3)  # the trellis decoders converge in fewer, cheaper iterations
    out = subprocess.check_output([ref, "time", str(snr), str(nframes), str(method)], cwd=os.path.join(ROOT, "data", "nb")).decode()
    j = json.loads(out.strip().splitlines()[-1])
    return {"value": j["frames_per_s"], "unit": "codewords/s", "cores": 1, "kind": "reference",
            "sample": "first %d frames of the same seed-173 stream at Eb/N0 %.1f dB, reference %s (maxIT 20, early exit), "
                      "mean %.2f iterations, %.1f s" % (nframes, snr, {0: "Decoding_EMS", 1: "Decoding_TMM", 3: "Decoding_layered_TMM"}[method],
                                                         j["mean_iters"], j["seconds"])}


def run_nb(args, rank, world, dev, dist):
    from cuda_ldpc_amd import nbldpc as nb
    name, q, _, method, frames, snr, iters = WORKLOADS[args.workload]
    q = q or 64
    if args.frames:
        frames = args.frames
    if args.snr is not None:
        snr = args.snr
    nbd = os.path.join(ROOT, "data", "nb")
    mul, _, _ = nb.GFInitial(q, os.path.join(nbd, "GF", "Arith.Table.GF.%d.txt" % q))
    code = nb.NBCode(os.path.join(nbd, name), mul)
    cw = np.loadtxt(os.path.join(nbd, "codeword_bds_gf64.txt"), dtype=np.int32) if q == 64 else np.zeros(code.N, np.int32)
    block = min(1024, frames)
    seed = np.array([173, 173, 173], np.int32)
    sigma = nb.sigma_of(snr, code.rate)
    rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw) for _ in range(block)])
    reps = (frames + block - 1) // block
    rxt = torch.from_numpy(rx).to(dev).repeat(reps, 1)[:frames].contiguous()
    Lch = nb.Demodulate(code, rxt, sigma)  # resident in HBM before the timed region
    cwd = torch.from_numpy(cw).to(dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def decode():
        if method == 0:
            return nb.Decoding_EMS(code, Lch, 2, 2, iters, stream=stream)
        return nb.Decoding_TMM(code, Lch, iters, layered=(method == 3), stream=stream)

    for _ in range(args.warmup):
        nb.Statistic(code, counters, decode(), cwd, stream=stream)
    counters.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        r = decode()
        ev[k][1].record(stream)
        nb.Statistic(code, counters, r, cwd, stream=stream)
    tot = counters.clone()
    if world > 1:
        dist.all_reduce(tot)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    if rank != 0:
        return
    n_all = frames * world * args.steps
    alg_bytes = (4 * code.N * (code.q - 1) + 4 * code.N) * frames  # L_ch in + symbols out
    kname = {0: "k_nb_ems2<64>" if q == 64 else "k_nb_ems_wide<%d>" % q, 1: "k_nb_tmm<64, false>", 3: "k_nb_tmm<64, true>"}[method]
    mname = {0: "GF(%d) EMS" % q, 1: "GF(64) trellis min-max", 3: "GF(64) layered trellis min-max"}[method]
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    c = tot.cpu().tolist()
    out = {
        "metric": "decoded codewords/sec (%s, maxIT %d, per-frame syndrome exit as the reference)" % (mname, iters),
        "value": n_all / elapsed, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "myNBLDPC %s %s batch=%d codewords/GPU Eb/N0=%.1fdB" % ("BDS N576_K288" if q == 64 else name.replace(".txt", ""),
                                                                                     ("GF(%d) EMS(Nm=2,Nc=2)" % q) if method == 0 else mname, frames, snr),
                   "kernel": "%s %s" % (kname, "two frames in flight per workgroup (walking waves + sorting waves)" if kname.startswith("k_nb_ems2") else "one frame per workgroup"), "frames_per_gpu": frames, "sharding": "frames, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(kname)[0], "traffic_source": pmc_traffic(kname)[1], "kernel": kname, "kernel_ms": kern_ms,
                     # (a frame occupies one half of a workgroup's pipeline for mean_iterations + P(converged) half-steps)
                     "onchip": pmc_onchip(kname, kern_ms, frames=frames, iters=c[2] / n_all + 1.0 - c[0] / n_all),
                     "algorithmic_bytes_per_launch": alg_bytes},
        "stats": {"frames": n_all, "error_frames": c[0], "symbol_errors": c[1], "FER": c[0] / n_all, "SER": c[1] / n_all / code.N,
                  "mean_iterations": c[2] / n_all},
    }
    if world == 1 and not args.no_cpu_baseline:
        cb = nb_cpu_baseline(snr, args.cpu_frames, method, q, name)
        if cb:
            out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0  # rehearsal of the N>1 code path on a one-GPU box (use with BENCH_DIST_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import cuda_ldpc_amd as C  # raises if the HIP extension is missing: no fallback

    if args.workload.startswith("NB_"):
        run_nb(args, rank, world, dev, dist if world > 1 else None)
        if world > 1:
            dist.destroy_process_group()
        return

    name, J, L, Z, frames, snr, iters = WORKLOADS[args.workload]
    if args.frames:
        frames = args.frames
    if args.iters:
        iters = args.iters
    code = C.BinaryCode.from_blockh(os.path.join(ROOT, "data", "bldpc", name), J, L, Z)
    kernel = {"auto": C.KERNEL_AUTO, "qc": C.KERNEL_QC_LDS, "table": C.KERNEL_TABLE}[args.kernel]
    N = code.N

    # synthetic input: reference AWGN stream, one 4096-frame block tiled to the batch (host generation is
    # serial: 2 LCG draws + 3 libm calls per sample, SURVEY 7 "hard parts")
    block = min(4096, frames)
    seed = np.array([173, 173, 173], np.int32)
    y_block = C.AWGNChannel_CPU(seed, C.sigma_of(snr), N, block)  # [N, block]
    reps = (frames + block - 1) // block
    y = torch.from_numpy(y_block).to(dev).repeat(1, reps)[:, :frames].contiguous()
    D = torch.empty((N + 1, frames), dtype=torch.int32, device=dev)
    SIM = C.SimCounters()
    SIM._dev = torch.zeros(5, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    import ctypes
    from cuda_ldpc_amd._lib import check, lib

    def step():  # LDPC_Decoder_GPU + Statistic, the pair of calls of the reference's Simulation_GPU loop (Simulation.cu:143-145)
        C.Decode_Statistic(code, y, SIM._dev, max_iter=iters, exit_mode=C.EXIT_FIXED, kernel=kernel, D=D, stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    code.set_profiling(True)
    for _ in range(args.warmup):
        step()
    SIM._dev.zero_()
    torch.cuda.synchronize(dev)
    code.kernel_ms_mean()  # drop the warm-up pairs
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)  # HIP events on the stream the kernel is launched on
        C.Decode_Statistic(code, y, SIM._dev, max_iter=iters, exit_mode=C.EXIT_FIXED, kernel=kernel, D=D, stream=stream)
        ev[k][1].record(stream)
    counters = SIM._dev.clone()
    if world > 1:
        dist.all_reduce(counters)  # the only collective: 5 int64 error counters
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    call_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps  # whole bldpc_decode call (regroup + decode + unpack)
    # dominant kernel alone: every timed step recorded its own event pair inside bldpc_decode (on the kernel's stream); the mean
    # over ALL timed steps (the library keeps the last 64 pairs), queried after the timed region so that nothing synchronises in it
    kern_ms, kern_n = code.kernel_ms_mean()

    if rank == 0:
        total_cw = frames * world * args.steps
        alg_bytes = (8 * N + 4) * frames  # fp32 LLR in + int32 hard bits out + flag, per launch
        model = "io: (8N+4) B per codeword"
        if code.last_kernel.startswith("table"):
            # the table kernels keep Memory_RQ in HBM: their algorithmic traffic is the reference schedule's
            # streaming model, 8N + iters*(16E+8N) bytes per codeword (secondary figure; E = nnz*Z edges);
            # the timed "kernel" is the whole VN/CN launch sequence of the call
            E = code.nnz * code.Z
            alg_bytes = (8 * N + (iters - 1) * (16 * E + 8 * N) + (8 * E + 8 * N)) * frames  # last iteration has no CN pass
            model = "streaming: 8N + iters*(16E+8N) B per codeword (messages resident in HBM)"
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        c = counters.cpu().tolist()
        out = {
            "metric": "decoded codewords/sec (50-iter flooding min-sum, fixed iterations)",
            "value": total_cw / elapsed, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s rate-%.3f batch=%d codewords/GPU %d iters Es/N0=%.1fdB" % (args.workload, code.K / N, frames, iters, snr),
                       "kernel": code.last_kernel, "frames_per_gpu": frames, "sharding": "frames, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(code.last_kernel)[0], "traffic_source": pmc_traffic(code.last_kernel)[1],
                         "kernel": code.last_kernel, "kernel_ms": kern_ms, "kernel_ms_launches_averaged": kern_n, "decode_call_ms": call_ms,
                         "achieved_decode_call": alg_bytes / (call_ms * 1e-3) / 1e9,  # the whole bldpc_decode_statistic call: (regroup +) decode + unpack with the error counts + the counters
                         "algorithmic_bytes_per_launch": alg_bytes, "model": model,
                         "onchip": pmc_onchip(code.last_kernel, kern_ms, frames=frames, iters=iters)},
            "stats": {"frames": frames * world * args.steps, "error_frames": c[0], "error_bits": c[1],
                      "FER": c[0] / (frames * world * args.steps), "BER": c[1] / (frames * world * args.steps) / code.K},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, J, L, Z, snr, iters, y_block, args.cpu_frames)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
