"""Host-side mirror of the reference's non-binary LDPC entry points (myNBLDPC/).

Same names and argument meaning as src/Simulation.cpp, src/GF.cpp, src/LDPC_Decoder.cpp and
src/LDPC_Encoder.cpp; the `define.h` macros (GFQ, maxdc, EMS_NM, EMS_NC, maxIT) become arguments.
All computation goes through the C ABI of include/nbldpc.h; tensors are torch CUDA(HIP) tensors.
"""
import ctypes

import numpy as np
import torch

from ._lib import LdpcError, lib

c_int, c_void_p, c_char_p, c_float = ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_float

lib.nbldpc_last_error.restype = c_char_p
lib.nbldpc_read_matrix.argtypes = [c_char_p] + [c_void_p] * 7
lib.nbldpc_gf_load.argtypes = [c_char_p, c_int, c_void_p, c_void_p, c_void_p]
lib.nbldpc_gf_generate.argtypes = [c_int, ctypes.c_uint, c_void_p, c_void_p, c_void_p]
lib.nbldpc_code_create.argtypes = [c_int] * 5 + [c_void_p] * 7 + [ctypes.POINTER(c_void_p)]
lib.nbldpc_code_destroy.argtypes = [c_void_p]
lib.nbldpc_ems_decode_batch.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 6
lib.nbldpc_tmm_decode_batch.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 6
lib.nbldpc_demodulate_bpsk.argtypes = [c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p]
lib.nbldpc_statistic.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
lib.nbldpc_awgn_channel_host.argtypes = [c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]
lib.nbldpc_awgn_channel_device.argtypes = [c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
lib.nbldpc_read_constellation.argtypes = [c_char_p, c_int, c_void_p]
lib.nbldpc_last_kernel.argtypes = [c_void_p]
lib.nbldpc_last_kernel.restype = ctypes.c_char_p
lib.nbldpc_awgn_channel_host_qam.argtypes = [c_void_p, c_float, c_void_p, c_int, c_void_p, c_int, c_void_p]
lib.nbldpc_awgn_channel_device_qam.argtypes = [c_void_p, c_float, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]
lib.nbldpc_demodulate_qam.argtypes = [c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p]
lib.nbldpc_sigma.restype = c_float
lib.nbldpc_sigma.argtypes = [c_float, c_int, c_int, c_float]


def _check(rc, what):
    if rc != 0:
        raise LdpcError("%s failed (%d): %s" % (what, rc, lib.nbldpc_last_error().decode(errors="replace")))


def _np(a):
    return a.ctypes.data_as(c_void_p)


def _dev(t):
    return None if t is None else c_void_p(t.data_ptr())


def GFInitial(q, path=None, primitive_poly=None):
    """GFInitial (GF.cpp:68-117) -> (TableMultiply[q,q], TableAdd[q,q], TableInverse[q]) uint32 host arrays.
    Either read the reference's table file or generate from the primitive polynomial."""
    mul = np.zeros((q, q), np.uint32)
    add = np.zeros((q, q), np.uint32)
    inv = np.zeros(q, np.uint32)
    if path is not None:
        _check(lib.nbldpc_gf_load(str(path).encode(), q, _np(mul), _np(add), _np(inv)), "GFInitial")
    else:
        _check(lib.nbldpc_gf_generate(q, primitive_poly, _np(mul), _np(add), _np(inv)), "GFInitial(generate)")
    return mul, add, inv


class NBCode:
    """Get_H (Simulation.cpp:347-467) + device upload. Holds the flattened VN/CN arrays on the host too."""

    def __init__(self, matrix_path, TableMultiply):
        dims = np.zeros(5, np.int32)
        _check(lib.nbldpc_read_matrix(str(matrix_path).encode(), _np(dims), None, None, None, None, None, None), "Get_H")
        self.N, self.M, self.q, self.dv, self.dc = (int(x) for x in dims)
        self.m = int(np.log2(self.q))
        self.vn_weight = np.zeros(self.N, np.int32)
        self.vn_linkCNs = np.zeros((self.N, self.dv), np.int32)
        self.vn_linkCNs_GF = np.zeros((self.N, self.dv), np.int32)
        self.cn_weight = np.zeros(self.M, np.int32)
        self.cn_linkVNs = np.zeros((self.M, self.dc), np.int32)
        self.cn_linkVNs_GF = np.zeros((self.M, self.dc), np.int32)
        _check(lib.nbldpc_read_matrix(str(matrix_path).encode(), _np(dims), _np(self.vn_weight), _np(self.vn_linkCNs),
                                      _np(self.vn_linkCNs_GF), _np(self.cn_weight), _np(self.cn_linkVNs), _np(self.cn_linkVNs_GF)),
               "Get_H")
        self.rate = np.float32(self.N - self.M) / np.float32(self.N)  # H->rate (Simulation.cpp:365)
        self.TableMultiply = np.ascontiguousarray(TableMultiply, np.uint32)
        h = c_void_p()
        _check(lib.nbldpc_code_create(self.N, self.M, self.q, self.dv, self.dc, _np(self.vn_weight), _np(self.vn_linkCNs),
                                      _np(self.vn_linkCNs_GF), _np(self.cn_weight), _np(self.cn_linkVNs), _np(self.cn_linkVNs_GF),
                                      _np(self.TableMultiply), ctypes.byref(h)), "nbldpc_code_create")
        self._h = h

    @property
    def last_kernel(self):
        """Name of the kernel the last decode call on this code launched (nbldpc_last_kernel)."""
        return lib.nbldpc_last_kernel(self._h).decode()

    def close(self):
        if getattr(self, "_h", None):
            lib.nbldpc_code_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def Decoding_EMS(code, L_ch, EMS_Nm=2, EMS_Nc=2, maxIT=20, maxdc=0, want_state=False, stream=None):
    """Decoding_EMS (LDPC_Decoder.cpp:172-317) for a batch.

    L_ch: CUDA float32 [B, N, q-1].  Returns dict(DecodeOutput [B,N] int32, iter_number [B], ok [B],
    LLR [B,N,q-1] | None, L_c2v [B,M,dc,q-1] | None), all on the device.  want_state: True = both state arrays,
    "llr" = VN[].LLR only (the two-frame pipeline kernel does not produce CN[].L_c2v: a call that asks for it runs on k_nb_ems)."""
    if not (L_ch.is_cuda and L_ch.dtype == torch.float32 and L_ch.is_contiguous()):
        raise ValueError("L_ch must be a contiguous CUDA float32 tensor")
    if L_ch.dim() != 3 or L_ch.shape[1] != code.N or L_ch.shape[2] != code.q - 1:
        raise ValueError("L_ch must be [B, N=%d, q-1=%d]" % (code.N, code.q - 1))
    B, dev = int(L_ch.shape[0]), L_ch.device
    out = torch.empty((B, code.N), dtype=torch.int32, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    ok = torch.empty(B, dtype=torch.int32, device=dev)
    LLR = torch.empty((B, code.N, code.q - 1), dtype=torch.float32, device=dev) if want_state else None
    c2v = torch.empty((B, code.M, code.dc, code.q - 1), dtype=torch.float32, device=dev) if (want_state and want_state != "llr") else None
    st = c_void_p((stream or torch.cuda.current_stream(dev)).cuda_stream)
    _check(lib.nbldpc_ems_decode_batch(code._h, _dev(L_ch), B, EMS_Nm, EMS_Nc, maxIT, maxdc, _dev(out), _dev(iters), _dev(ok),
                                       _dev(LLR), _dev(c2v), st), "Decoding_EMS")
    return dict(DecodeOutput=out, iter_number=iters, ok=ok, LLR=LLR, L_c2v=c2v)


def Decoding_TMM(code, L_ch, maxIT=20, layered=False, want_state=False, stream=None):
    """Decoding_TMM (LDPC_Decoder.cpp:361-558) / Decoding_layered_TMM (:560-702, layered=True) for a batch.
    Returns the same dict as Decoding_EMS; LLR [B,N,q] and L_c2v [B,M,dc,q] carry all q entries per vector."""
    if not (L_ch.is_cuda and L_ch.dtype == torch.float32 and L_ch.is_contiguous()):
        raise ValueError("L_ch must be a contiguous CUDA float32 tensor")
    if L_ch.dim() != 3 or L_ch.shape[1] != code.N or L_ch.shape[2] != code.q - 1:
        raise ValueError("L_ch must be [B, N=%d, q-1=%d]" % (code.N, code.q - 1))
    B, dev = int(L_ch.shape[0]), L_ch.device
    out = torch.empty((B, code.N), dtype=torch.int32, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    ok = torch.empty(B, dtype=torch.int32, device=dev)
    LLR = torch.empty((B, code.N, code.q), dtype=torch.float32, device=dev) if want_state else None
    c2v = torch.empty((B, code.M, code.dc, code.q), dtype=torch.float32, device=dev) if want_state else None
    st = c_void_p((stream or torch.cuda.current_stream(dev)).cuda_stream)
    _check(lib.nbldpc_tmm_decode_batch(code._h, _dev(L_ch), B, 1 if layered else 0, maxIT, _dev(out), _dev(iters), _dev(ok), _dev(LLR),
                                       _dev(c2v), st), "Decoding_TMM")
    return dict(DecodeOutput=out, iter_number=iters, ok=ok, LLR=LLR, L_c2v=c2v)


def Get_CONSTELLATION(path, n_QAM):
    """Get_CONSTELLATION (Simulation.cpp:313-338) -> host float32 [n_QAM, 2] (Real, Image)."""
    con = np.zeros((n_QAM, 2), np.float32)
    _check(lib.nbldpc_read_constellation(str(path).encode(), n_QAM, _np(con)), "Get_CONSTELLATION")
    return con


def Demodulate(code, rx, sigma, stream=None, CONSTELLATION=None):
    """Demodulate on the device.  BPSK branch (LDPC_Decoder.cpp:132-157): rx [B, N*m] -> L_ch [B, N, q-1].  With
    CONSTELLATION (CUDA float32 [q, 2]) the n_QAM != 2 branch (:160-169): rx [B, N, 2] (include/nbldpc.h)."""
    if CONSTELLATION is not None:
        if not (rx.is_cuda and rx.dtype == torch.float32 and rx.is_contiguous() and rx.dim() == 3 and tuple(rx.shape[1:]) == (code.N, 2)):
            raise ValueError("rx must be a contiguous CUDA float32 tensor [B, N, 2]")
        if not (CONSTELLATION.is_cuda and CONSTELLATION.dtype == torch.float32 and CONSTELLATION.is_contiguous()
                and tuple(CONSTELLATION.shape) == (code.q, 2)):
            raise ValueError("CONSTELLATION must be a contiguous CUDA float32 tensor [q, 2]")
        B = int(rx.shape[0])
        Lch = torch.empty((B, code.N, code.q - 1), dtype=torch.float32, device=rx.device)
        st = c_void_p((stream or torch.cuda.current_stream(rx.device)).cuda_stream)
        _check(lib.nbldpc_demodulate_qam(code._h, _dev(rx), _dev(CONSTELLATION), c_float(sigma), B, _dev(Lch), st), "Demodulate")
        return Lch
    if not (rx.is_cuda and rx.dtype == torch.float32 and rx.is_contiguous() and rx.dim() == 2 and rx.shape[1] == code.N * code.m):
        raise ValueError("rx must be a contiguous CUDA float32 tensor [B, N*m]")
    B = int(rx.shape[0])
    Lch = torch.empty((B, code.N, code.q - 1), dtype=torch.float32, device=rx.device)
    st = c_void_p((stream or torch.cuda.current_stream(rx.device)).cuda_stream)
    _check(lib.nbldpc_demodulate_bpsk(code._h, _dev(rx), c_float(sigma), B, _dev(Lch), st), "Demodulate")
    return Lch


def AWGNChannel_CPU(seed, sigma, code, CodeWord_sym, CONSTELLATION=None):
    """Modulate + AWGNChannel_CPU (LDPC_Encoder.cpp:18-68) for ONE frame; seed advanced.  BPSK -> host rx [N*m]; with
    CONSTELLATION (host float32 [q, 2]) the n_QAM != 2 branch -> host rx [N, 2]."""
    if not (isinstance(seed, np.ndarray) and seed.dtype == np.int32 and seed.size == 3):
        raise ValueError("seed must be an int32 numpy array of 3")
    cw = np.ascontiguousarray(CodeWord_sym, np.int32)
    if CONSTELLATION is not None:
        con = np.ascontiguousarray(CONSTELLATION, np.float32)
        rx = np.empty((code.N, 2), np.float32)
        _check(lib.nbldpc_awgn_channel_host_qam(_np(seed), c_float(sigma), _np(cw), code.N, _np(con), con.shape[0], _np(rx)), "AWGNChannel_CPU")
        return rx
    rx = np.empty(code.N * code.m, np.float32)
    _check(lib.nbldpc_awgn_channel_host(_np(seed), c_float(sigma), _np(cw), code.N, code.m, _np(rx)), "AWGNChannel_CPU")
    return rx


def AWGNChannel_GPU(seed, sigma, code, CodeWord_sym_dev, B, stream=None, CONSTELLATION=None):
    """Device-side Modulate + AWGNChannel for B consecutive frames of the same stream (LCG jump-ahead: the uniforms are
    the reference's, the samples may differ from the host libm's by an ulp).  Returns rx CUDA float32 [B, N*m]; seed
    advanced exactly like B calls of AWGNChannel_CPU."""
    if not (isinstance(seed, np.ndarray) and seed.dtype == np.int32 and seed.size == 3):
        raise ValueError("seed must be an int32 numpy array of 3")
    if not (CodeWord_sym_dev.is_cuda and CodeWord_sym_dev.dtype == torch.int32 and CodeWord_sym_dev.numel() == code.N):
        raise ValueError("CodeWord_sym_dev must be a CUDA int32 tensor of N symbols")
    if CONSTELLATION is not None:  # n_QAM != 2 branch: rx [B, N, 2], four draws per SYMBOL
        if not (CONSTELLATION.is_cuda and CONSTELLATION.dtype == torch.float32 and CONSTELLATION.is_contiguous()
                and tuple(CONSTELLATION.shape) == (code.q, 2)):
            raise ValueError("CONSTELLATION must be a contiguous CUDA float32 tensor [q, 2]")
        rx = torch.empty((B, code.N, 2), dtype=torch.float32, device=CodeWord_sym_dev.device)
        st = c_void_p((stream or torch.cuda.current_stream(rx.device)).cuda_stream)
        _check(lib.nbldpc_awgn_channel_device_qam(_np(seed), c_float(sigma), _dev(CodeWord_sym_dev), code.N, _dev(CONSTELLATION), B, _dev(rx), st),
               "AWGNChannel_GPU")
        return rx
    rx = torch.empty((B, code.N * code.m), dtype=torch.float32, device=CodeWord_sym_dev.device)
    st = c_void_p((stream or torch.cuda.current_stream(rx.device)).cuda_stream)
    _check(lib.nbldpc_awgn_channel_device(_np(seed), c_float(sigma), _dev(CodeWord_sym_dev), code.N, code.m, B, _dev(rx), st), "AWGNChannel_GPU")
    return rx


def seed_after(seed, frames, code, qam=False):
    """The RandomModule state after `frames` more frames (4 draws per bit; per symbol with a QAM constellation)."""
    a, m = (249, 251, 252), (61967, 63443, 63599)
    k = 4 * code.N * (1 if qam else code.m) * int(frames)
    return np.array([(int(seed[i]) * pow(a[i], k, m[i])) % m[i] for i in range(3)], np.int32)


def sigma_of(SNR, rate, snrtype=0, n_QAM=2):
    """sigma of a sweep point (main.cu:221-228); snrtype 0 = Eb/N0 (reference default, define.h:45)."""
    return float(lib.nbldpc_sigma(np.float32(SNR), snrtype, n_QAM, np.float32(rate)))


def Statistic(code, counters, res, CodeWord_sym_dev, stream=None):
    """Statistic (Simulation.cpp:256-279) on the device; counters: CUDA int64[4], accumulated."""
    B = int(res["DecodeOutput"].shape[0])
    st = c_void_p((stream or torch.cuda.current_stream(counters.device)).cuda_stream)
    _check(lib.nbldpc_statistic(code._h, _dev(res["DecodeOutput"]), _dev(res["iter_number"]), _dev(res["ok"]), _dev(CodeWord_sym_dev),
                                B, _dev(counters), st), "Statistic")
