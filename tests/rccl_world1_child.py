"""Child process of test_rccl_world_size_one_smoke: ONE rank started by `python -m torch.distributed.run --nproc-per-node 1`.

Initialises the process group with backend "nccl" (= RCCL on ROCm) on cuda:0 before anything else touches the GPU, runs
Simulation_GPU of the binary program for two batches with dist=dist -- so that sharding.allreduce_counters issues the
all-reduce of the five int64 error counters through librccl -- then the NB statistics through the same collective, and prints
one JSON line with the counters.  World size 1: this proves that RCCL loads and that the collective of the multi-GPU path
executes on MI355X; it is not a scaling measurement.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    import cuda_ldpc_amd as C
    from cuda_ldpc_amd import sharding
    from cuda_ldpc_amd.simulation import Simulation_GPU
    J, L, Z = 4, 24, 96
    code = C.BinaryCode.from_blockh(os.path.join(ROOT, "data", "bldpc", "J4_L24_Z96_BlockH.txt"), J, L, Z)
    seed = np.array([173, 173, 173], np.int32)
    SIM = C.SimCounters()
    Simulation_GPU(code, seed, C.sigma_of(3.0), SIM, Num_Frames_OneTime=2048, maxIT=50, exit_mode=C.EXIT_PER_FRAME, dist=dist, device=dev,
                   max_batches=2, log=None, device_channel=True)
    probe = torch.arange(5, dtype=torch.int64, device=dev)
    sharding.allreduce_counters(probe, dist)  # world size 1: the sum over ranks is the tensor itself
    torch.cuda.synchronize(dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "frames": SIM.num_Frames, "error_frames": SIM.num_Error_Frames,
           "error_bits": SIM.num_Error_Bits, "total_iteration": SIM.Total_Iteration, "seed": seed.tolist(), "probe": probe.cpu().tolist(),
           "rccl_loaded": any("librccl" in ln for ln in open("/proc/self/maps"))}
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
