"""Frame sharding across ranks (one process per GPU) -- host logic only, no GPU needed.

Codewords are independent, so the batch of one Simulation step is cut into contiguous frame ranges, one
per rank, with NO data-path collective; the only exchange is the all-reduce (sum) of the error counters
(RCCL on GPUs = torch.distributed backend "nccl"; "gloo" in the CPU tests).

The reference draws its channel noise from ONE serial stream (three LCGs, LDPC_Encoder.cu:45-56), frame
after frame.  An LCG can be advanced k steps at once (seed * a^k mod m), so each rank jumps straight to
the first draw of its own frame range and generates exactly the samples a single process would have
generated for those frames.
"""
import numpy as np

LCG_A = (249, 251, 252)        # RandomModule multipliers (LDPC_Encoder.cu:48-50)
LCG_M = (61967, 63443, 63599)  # and moduli


def shard_frames(F, world, rank):
    """Contiguous range of a batch of F frames owned by `rank`: (first, count). Earlier ranks take the remainder."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank %d / world %d" % (rank, world))
    base, rem = divmod(F, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def lcg_jump(seed, draws):
    """State of the three LCGs after `draws` calls of RandomModule (each call advances all three once)."""
    return np.array([(int(seed[i]) * pow(LCG_A[i], int(draws), LCG_M[i])) % LCG_M[i] for i in range(3)], np.int32)


def binary_draws_per_frame(N):
    return 2 * N   # u1, u2 per sample (LDPC_Encoder.cu:32-33)


def nb_draws_per_frame(N, m):
    return 4 * N * m  # Real and Image part, two draws each (LDPC_Encoder.cpp:57-66)


def allreduce_counters(counters, dist=None):
    """Sum an int64 counter tensor over all ranks (no-op without a process group). Returns the tensor.
    A caller that passes an initialised process group gets the collective at every world size, 1 included (40 bytes: the
    one-GPU RCCL smoke test exercises exactly this call)."""
    if dist is not None and dist.is_initialized():
        dist.all_reduce(counters)
    return counters
