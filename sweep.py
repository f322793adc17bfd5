#!/usr/bin/env python3
"""BER/FER sweeps in the reference's output format, driven through the C ABI.

    python sweep.py binary --matrix data/bldpc/J4_L24_Z96_BlockH.txt --J 4 --L 24 --Z 96 --start 0 --stop 4.4 --step 0.2
    python sweep.py nb     --start 0 --stop 5 --step 0.5
    python -m torch.distributed.run --nproc-per-node N sweep.py binary ...      (frames sharded over N GPUs)

binary: rows `SNR NTF NEF FER BER AverIT FER_F FER_A` as Simulation.cu:272 prints them (Es/N0, seeds 173/173/173,
batches of --batch frames, stop at >= 50 error frames and >= 10000 frames, or --max-batches).
nb:     rows `SNR NTF NEF FER BER AverIT` as Simulation.cpp:198 (Eb/N0, stop at >= 50 error frames and >= 1000 frames).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["binary", "nb"])
    ap.add_argument("--matrix", default=os.path.join(ROOT, "data", "bldpc", "J4_L24_Z96_BlockH.txt"))
    ap.add_argument("--J", type=int, default=4)
    ap.add_argument("--L", type=int, default=24)
    ap.add_argument("--Z", type=int, default=96)
    ap.add_argument("--start", type=float, default=0.0)
    ap.add_argument("--stop", type=float, default=4.4)
    ap.add_argument("--step", type=float, default=0.2)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--max-batches", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--fixed", action="store_true", help="fixed iteration count instead of the reference's batch-global early exit")
    ap.add_argument("--per-frame", action="store_true", help="binary: every frame stops on its own flag (the reference rule with Num_Frames_OneTime = 1)")
    ap.add_argument("--qam", type=int, default=2, choices=[2, 64], help="nb: n_QAM (define.h:25): 2 = BPSK, 64 = Constellation/GRAY_64QAM.txt, one point per GF(64) symbol")
    ap.add_argument("--method", type=int, default=0, choices=[0, 1, 2, 3], help="NB decoder_method (define.h:37): 0 EMS, 1 TMM, 2 log-QSPA, 3 layered TMM")
    ap.add_argument("--device-channel", action="store_true", help="generate the AWGN samples on the GPU (same RNG draws, device libm)")
    ap.add_argument("--as-written", action="store_true", help="decode on the reference's Transform_H table as written (SURVEY F3)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank = int(os.environ.get("RANK", "0"))
    import cuda_ldpc_amd as C

    if args.which == "binary":
        from cuda_ldpc_amd.simulation import sweep
        if args.as_written:
            H, wc, wv = C.Get_H(args.matrix, args.J, args.L)
            code = C.BinaryCode.from_table(args.J, args.L, args.Z, wc, wv, C.Transform_H(H, args.J, args.L, args.Z, wc, wv, as_written=True))
        else:
            code = C.BinaryCode.from_blockh(args.matrix, args.J, args.L, args.Z)
        if rank == 0:
            print("# %s N=%d K=%d, %s, maxIT=%d, batch=%d x %d GPU(s)" % (os.path.basename(args.matrix), code.N, code.K,
                  "fixed iterations" if args.fixed else ("per-frame early exit" if args.per_frame else "batch-global early exit"),
                  args.iters or 50, args.batch, world))
            print("# SNR      NTF   NEF         FER         BER  AverIT       FER_F      FER_A")
        sweep(code, args.start, args.stop, args.step, snrtype=1, dist=dist, Num_Frames_OneTime=args.batch, maxIT=args.iters or 50,
              exit_mode=C.EXIT_FIXED if args.fixed else (C.EXIT_PER_FRAME if args.per_frame else C.EXIT_BATCH_GLOBAL), max_batches=args.max_batches, displayStep=10 ** 12, device_channel=args.device_channel,
              log=print if rank == 0 else None)
    else:
        from cuda_ldpc_amd import nbldpc as nb
        from cuda_ldpc_amd.nb_simulation import sweep
        nbd = os.path.join(ROOT, "data", "nb")
        mul, _, _ = nb.GFInitial(64, os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
        code = nb.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), mul)
        cw = np.loadtxt(os.path.join(nbd, "codeword_bds_gf64.txt"), dtype=np.int32)
        con = None if args.qam == 2 else nb.Get_CONSTELLATION(os.path.join(nbd, "Constellation", "GRAY_64QAM.txt"), args.qam)
        print("# BDS.576.288.GF.64 N=%d symbols GF(%d), %s, maxIT=%d, %s" % (code.N, code.q, ["EMS(2,2)", "trellis min-max", "log-QSPA = EMS(q,dc-1)", "layered trellis min-max"][args.method], args.iters or 20,
              "BPSK" if args.qam == 2 else "%d-QAM (Gray), one point per symbol" % args.qam))
        print("# SNR      NTF   NEF         FER         BER  AverIT")
        nbatch = args.batch if args.device_channel else min(args.batch, 1024)  # the host channel is serial: keep its batches small
        sweep(code, cw, args.start, args.stop, args.step, maxIT=args.iters or 20, batch=nbatch,
              max_frames=None if args.max_batches is None else args.max_batches * nbatch, device_channel=args.device_channel,
              decoder_method=args.method, n_QAM=args.qam, CONSTELLATION=con)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
