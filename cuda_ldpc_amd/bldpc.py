"""Host-side mirror of the reference's binary LDPC entry points.

Same names and argument meaning as bldpc_实习/{Simulation,LDPC_Decoder}.cu; the
compile-time macros of define.cuh (J, L, Z, Num_Frames_OneTime, maxIT, msgLen)
become run-time arguments.  Everything computes through the C ABI of
include/bldpc.h on the GPU; tensors are torch CUDA(HIP) tensors.
"""
import ctypes
from dataclasses import dataclass, field

import numpy as np
import torch

from ._lib import check, lib

EXIT_FIXED, EXIT_BATCH_GLOBAL, EXIT_PER_FRAME = 0, 1, 2
KERNEL_AUTO, KERNEL_TABLE, KERNEL_QC_LDS = 0, 1, 2


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dev_ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check_D(D, N, F, dev):
    """A caller-supplied D goes to the C ABI as a raw pointer: a wrong shape is an out-of-bounds device write and a
    misaligned one faults in the 16-byte stores of the unpack kernel -- refuse both here."""
    if not (torch.is_tensor(D) and D.is_cuda and D.device == dev and D.dtype == torch.int32 and D.is_contiguous()
            and D.dim() == 2 and tuple(D.shape) == (N + 1, F)):
        raise ValueError("D must be a contiguous CUDA int32 tensor [N+1=%d, F=%d] on %s" % (N + 1, F, dev))
    if D.data_ptr() % 16:
        raise ValueError("D must be 16-byte aligned")


def _check_counters(counters, n, dev):
    if not (torch.is_tensor(counters) and counters.is_cuda and counters.device == dev and counters.dtype == torch.int64
            and counters.is_contiguous() and counters.numel() >= n):
        raise ValueError("counters must be a contiguous CUDA int64 tensor of at least %d elements on %s" % (n, dev))


def Get_H(path, J, L):
    """Get_H (Simulation.cu:292-354): -> (H[J*L], Weight_Checknode[J+1], Weight_Variablenode[L+1]) int32 host arrays."""
    H = np.zeros(J * L, np.int32)
    wc = np.zeros(J + 1, np.int32)
    wv = np.zeros(L + 1, np.int32)
    check(lib.bldpc_read_blockh(str(path).encode(), J, L, _np_ptr(H), _np_ptr(wc), _np_ptr(wv)), "Get_H")
    return H, wc, wv


def Transform_H(H, J, L, Z, Weight_Checknode, Weight_Variablenode, as_written=False):
    """Transform_H (Simulation.cu:363-387): -> Address_Variablenode[N*Wv] int32 (host)."""
    H = np.ascontiguousarray(H, np.int32)
    wc = np.ascontiguousarray(Weight_Checknode, np.int32)
    wv = np.ascontiguousarray(Weight_Variablenode, np.int32)
    addr = np.zeros(L * Z * int(wv[L]), np.int32)
    check(lib.bldpc_transform_h(_np_ptr(H), J, L, Z, _np_ptr(wc), _np_ptr(wv), _np_ptr(addr), 1 if as_written else 0),
          "Transform_H")
    return addr


def AWGNChannel_CPU(seed, sigma, N, F, CodeWord=None):
    """AWGNChannel_CPU (LDPC_Encoder.cu:25-43). seed: int32[3] numpy array, advanced in place (AWGN->seed).
    Returns Channel_Out as a host float32 array [N, F] (frame-fastest)."""
    if not (isinstance(seed, np.ndarray) and seed.dtype == np.int32 and seed.size == 3):
        raise ValueError("seed must be an int32 numpy array of 3")
    out = np.empty((N, F), np.float32)
    cw = None if CodeWord is None else np.ascontiguousarray(CodeWord, np.int32)
    check(lib.bldpc_awgn_channel_host(_np_ptr(seed), ctypes.c_float(sigma), _np_ptr(out), None if cw is None else _np_ptr(cw), N, F),
          "AWGNChannel_CPU")
    return out


def AWGNChannel_GPU(seed, sigma, N, F, device=None, CodeWord=None, stream=None):
    """The same channel generated on the device (bldpc_awgn_channel_device): identical RandomModule draws via LCG
    jump-ahead, device libm for the Box-Muller transform.  Returns a CUDA float32 tensor [N, F]; seed advanced."""
    if not (isinstance(seed, np.ndarray) and seed.dtype == np.int32 and seed.size == 3):
        raise ValueError("seed must be an int32 numpy array of 3")
    device = device or torch.device("cuda", torch.cuda.current_device())
    out = torch.empty((N, F), dtype=torch.float32, device=device)
    st = ctypes.c_void_p((stream or torch.cuda.current_stream(device)).cuda_stream)
    check(lib.bldpc_awgn_channel_device(_np_ptr(seed), ctypes.c_float(sigma), _dev_ptr(out), _dev_ptr(CodeWord), N, F, st), "AWGNChannel_GPU")
    return out


def sigma_of(SNR, snrtype=1, rate=0.0):
    """sigma of a sweep point (main.cu:120-127); snrtype 1 = Es/N0 (the reference default, define.cuh:45)."""
    return float(lib.bldpc_sigma(np.float32(SNR), snrtype, np.float32(rate)))


class BinaryCode:
    """Device-resident code object (bldpc_code).  Build with from_blockh / from_shifts / from_table."""

    def __init__(self, handle, J, L, Z):
        self._h = handle
        self.J, self.L, self.Z = J, L, Z
        d = np.zeros(8, np.int32)
        check(lib.bldpc_code_dims(self._h, _np_ptr(d)), "bldpc_code_dims")
        self.N, self.M, self.K, self.Wc, self.Wv, self.nnz, self.levels, self.frames_per_wg = (int(x) for x in d)

    @classmethod
    def from_shifts(cls, H, J, L, Z):
        H = np.ascontiguousarray(H, np.int32)
        if H.size != J * L:
            raise ValueError("H must hold J*L shifts")
        h = ctypes.c_void_p()
        check(lib.bldpc_code_create_qc(J, L, Z, _np_ptr(H), ctypes.byref(h)), "bldpc_code_create_qc")
        return cls(h, J, L, Z)

    @classmethod
    def from_blockh(cls, path, J, L, Z):
        H, _, _ = Get_H(path, J, L)
        return cls.from_shifts(H, J, L, Z)

    @classmethod
    def from_table(cls, J, L, Z, Weight_Checknode, Weight_Variablenode, Address_Variablenode):
        wc = np.ascontiguousarray(Weight_Checknode, np.int32)
        wv = np.ascontiguousarray(Weight_Variablenode, np.int32)
        addr = np.ascontiguousarray(Address_Variablenode, np.int32)
        if wc.size != J + 1 or wv.size != L + 1 or addr.size != L * Z * int(wv[L]):
            raise ValueError("table shapes do not match J, L, Z")
        h = ctypes.c_void_p()
        check(lib.bldpc_code_create_table(J, L, Z, _np_ptr(wc), _np_ptr(wv), _np_ptr(addr), ctypes.byref(h)),
              "bldpc_code_create_table")
        return cls(h, J, L, Z)

    @property
    def last_kernel(self):
        return lib.bldpc_last_kernel(self._h).decode()

    def set_profiling(self, enable=True):
        check(lib.bldpc_set_profiling(self._h, 1 if enable else 0), "bldpc_set_profiling")

    def last_kernel_ms(self):
        """Elapsed ms of the dominant kernel of the last LDPC_Decoder_GPU call (HIP events on its stream)."""
        ms = ctypes.c_float(0)
        check(lib.bldpc_last_kernel_ms(self._h, ctypes.byref(ms)), "bldpc_last_kernel_ms")
        return ms.value

    def kernel_ms_mean(self):
        """(mean ms of the dominant kernel, calls averaged) over the profiled decode calls since the last query -- every call
        records its own event pair, nothing synchronises in between (bldpc_kernel_ms_mean)."""
        ms, n = ctypes.c_float(0), ctypes.c_int(0)
        check(lib.bldpc_kernel_ms_mean(self._h, ctypes.byref(ms), ctypes.byref(n)), "bldpc_kernel_ms_mean")
        return ms.value, n.value

    def close(self):
        if self._h:
            lib.bldpc_code_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def LDPC_Decoder_GPU(code, Channel_Out, max_iter=50, length=0, exit_mode=EXIT_BATCH_GLOBAL, kernel=KERNEL_AUTO,
                     D=None, want_app=False, want_flag_hist=False, stream=None):
    """LDPC_Decoder_GPU (LDPC_Decoder.cu:23-164).

    Channel_Out: CUDA float32 tensor [N, F] (frame-fastest, as the reference).
    Returns dict(D=int32 [N+1, F] on device, iteraTime=int, app=[N, F] or None, flag_hist=uint64-as-int64 [F] or None).

    exit_mode=EXIT_PER_FRAME (bldpc_decode_per_frame): every frame stops on its own flag, as the reference does with a
    batch of one frame; the result then carries iters=int32 [F] on the device (iterations per frame) and iteraTime is None.
    """
    if not (Channel_Out.is_cuda and Channel_Out.dtype == torch.float32 and Channel_Out.is_contiguous()):
        raise ValueError("Channel_Out must be a contiguous CUDA float32 tensor")
    if Channel_Out.dim() != 2 or Channel_Out.shape[0] != code.N:
        raise ValueError("Channel_Out must be [N=%d, F]" % code.N)
    F = int(Channel_Out.shape[1])
    dev = Channel_Out.device
    if D is None:
        D = torch.empty((code.N + 1, F), dtype=torch.int32, device=dev)
    else:
        _check_D(D, code.N, F, dev)
    app = torch.empty((code.N, F), dtype=torch.float32, device=dev) if want_app else None
    hist = torch.zeros(F, dtype=torch.int64, device=dev) if want_flag_hist else None
    it = ctypes.c_int(0)
    st = ctypes.c_void_p((stream or torch.cuda.current_stream(dev)).cuda_stream)
    if exit_mode == EXIT_PER_FRAME:
        if want_flag_hist:
            raise ValueError("flag history is not returned with per-frame exit (iters holds each frame's stop iteration)")
        iters = torch.empty(F, dtype=torch.int32, device=dev)
        check(lib.bldpc_decode_per_frame(code._h, _dev_ptr(Channel_Out), F, max_iter, length, kernel, _dev_ptr(D), _dev_ptr(app),
                                         _dev_ptr(iters), st), "LDPC_Decoder_GPU")
        return dict(D=D, iteraTime=None, app=app, flag_hist=None, iters=iters)
    check(lib.bldpc_decode(code._h, _dev_ptr(Channel_Out), F, max_iter, length, exit_mode, kernel, _dev_ptr(D), _dev_ptr(app),
                           _dev_ptr(hist), ctypes.byref(it), st), "LDPC_Decoder_GPU")
    return dict(D=D, iteraTime=it.value, app=app, flag_hist=hist)


def Decode_Statistic(code, Channel_Out, counters, max_iter=50, length=0, exit_mode=EXIT_BATCH_GLOBAL, kernel=KERNEL_AUTO, D=None, stream=None):
    """LDPC_Decoder_GPU followed by Statistic against the all-zero codeword, the pair of calls of Simulation_GPU's loop
    (Simulation.cu:143-145), as ONE call of the C ABI (bldpc_decode_statistic): same D, same iteration counts, same counters
    (device int64[5], accumulated); with a single-launch exit mode the errors are counted while D is written.
    Returns dict(D, iteraTime or None, iters or None)."""
    if not (Channel_Out.is_cuda and Channel_Out.dtype == torch.float32 and Channel_Out.is_contiguous()):
        raise ValueError("Channel_Out must be a contiguous CUDA float32 tensor")
    if Channel_Out.dim() != 2 or Channel_Out.shape[0] != code.N:
        raise ValueError("Channel_Out must be [N=%d, F]" % code.N)
    F = int(Channel_Out.shape[1])
    dev = Channel_Out.device
    _check_counters(counters, 5, dev)
    if D is None:
        D = torch.empty((code.N + 1, F), dtype=torch.int32, device=dev)
    else:
        _check_D(D, code.N, F, dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev) if exit_mode == EXIT_PER_FRAME else None
    it = ctypes.c_int(0)
    st = ctypes.c_void_p((stream or torch.cuda.current_stream(dev)).cuda_stream)
    check(lib.bldpc_decode_statistic(code._h, _dev_ptr(Channel_Out), F, max_iter, length, exit_mode, kernel, _dev_ptr(D), _dev_ptr(iters),
                                     _dev_ptr(counters), ctypes.byref(it), st), "Decode_Statistic")
    return dict(D=D, iteraTime=None if iters is not None else it.value, iters=iters)


@dataclass
class SimCounters:
    """The counters of struct Simulation (struct.cuh:17-33)."""
    num_Frames: int = 0
    num_Error_Frames: int = 0
    num_Error_Bits: int = 0
    Total_Iteration: int = 0
    num_False_Frames: int = 0
    num_Alarm_Frames: int = 0
    SNR: float = 0.0
    _dev: object = field(default=None, repr=False)

    def ratios(self, length):
        n = max(self.num_Frames, 1)
        return dict(FER=self.num_Error_Frames / n, BER=self.num_Error_Bits / n / length, AverageIT=self.Total_Iteration / n,
                    FER_False=self.num_False_Frames / n, FER_Alarm=self.num_Alarm_Frames / n)


def Statistic(SIM, code, D, iteraTime, length=0, CodeWord=None, leastErrorFrames=50, leastTestFrames=10000, stream=None):
    """Statistic (Simulation.cu:245-285) on the device; returns the reference's stop flag.

    The caller adds the batch to SIM.num_Frames first, like Simulation_GPU does (Simulation.cu:113).
    iteraTime: the batch's iteration count (int), or the per-frame counts of EXIT_PER_FRAME (CUDA int32 tensor [F])."""
    F = int(D.shape[1])
    if SIM._dev is None:
        SIM._dev = torch.zeros(5, dtype=torch.int64, device=D.device)
    st = ctypes.c_void_p((stream or torch.cuda.current_stream(D.device)).cuda_stream)
    if torch.is_tensor(iteraTime):
        check(lib.bldpc_statistic_per_frame(code._h, _dev_ptr(D), _dev_ptr(CodeWord), F, length, _dev_ptr(iteraTime), _dev_ptr(SIM._dev), st),
              "Statistic")
    else:
        check(lib.bldpc_statistic(code._h, _dev_ptr(D), _dev_ptr(CodeWord), F, length, iteraTime, _dev_ptr(SIM._dev), st), "Statistic")
    c = SIM._dev.cpu().tolist()
    SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration, SIM.num_False_Frames, SIM.num_Alarm_Frames = c
    return 1 if (SIM.num_Error_Frames >= leastErrorFrames and SIM.num_Frames >= leastTestFrames) else 0
