"""cuda_ldpc_amd -- MI355X-native LDPC belief-propagation decoders.

Python host-side mirror of the reference's entry points (gsw4869/CUDA_LDPC) over
the C ABI in include/*.h.  The compute lives in libcuda_ldpc_amd.so (hand-written
HIP for gfx950); PyTorch is used only for device memory, streams and
torch.distributed.  There is no CPU fallback: importing the decoders without
the built extension raises.
"""
from . import _lib  # noqa: F401
from .bldpc import (BinaryCode, Get_H, Transform_H, LDPC_Decoder_GPU, Decode_Statistic, Statistic, SimCounters, AWGNChannel_CPU, AWGNChannel_GPU, sigma_of,  # noqa: F401
                    EXIT_FIXED, EXIT_BATCH_GLOBAL, EXIT_PER_FRAME, KERNEL_AUTO, KERNEL_TABLE, KERNEL_QC_LDS)

__all__ = ["BinaryCode", "Get_H", "Transform_H", "LDPC_Decoder_GPU", "Decode_Statistic", "Statistic", "SimCounters", "AWGNChannel_CPU", "AWGNChannel_GPU", "sigma_of",
           "EXIT_FIXED", "EXIT_BATCH_GLOBAL", "EXIT_PER_FRAME", "KERNEL_AUTO", "KERNEL_TABLE", "KERNEL_QC_LDS"]
