"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``; the product package
``cuda_ldpc_amd`` must never import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


def build(force=False):
    """Compile liboracle.so (and, when /root/reference is present, _ref/nb_ref)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bldpc_oracle.c", "nbldpc_oracle.c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.orc_bldpc_sigma.restype = c_float
        L.orc_bldpc_sigma.argtypes = [c_float, c_int, c_float]
        L.orc_nb_sigma.restype = c_float
        L.orc_nb_sigma.argtypes = [c_float, c_int, c_int, c_float]
        L.orc_random_module.restype = c_float
        L.orc_fold_hash.restype = ctypes.c_uint32
        L.orc_fold_hash.argtypes = [c_void_p, c_size_t]
        L.orc_fold_hash_f32.restype = ctypes.c_uint32
        L.orc_fold_hash_f32.argtypes = [c_void_p, c_size_t]
        L.orc_bldpc_awgn.argtypes = [c_void_p, c_float, c_void_p, c_void_p, c_int, c_int]
        L.orc_bldpc_statistic.argtypes = [c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]
        L.orc_nb_awgn.argtypes = [c_void_p, c_float, c_void_p, c_void_p, c_int]
        L.orc_nb_demodulate_bpsk.argtypes = [c_void_p, c_float, c_int, c_int, c_int, c_void_p]
        L.orc_nb_read_constellation.argtypes = [ctypes.c_char_p, c_int, c_void_p]
        L.orc_nb_awgn_qam.argtypes = [c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p]
        L.orc_nb_demodulate_qam.argtypes = [c_void_p, c_void_p, c_float, c_int, c_int, c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def ref_binary():
    """Path of the reference NB decoder built from /root/reference (or None)."""
    p = os.path.join(_HERE, "_ref", "nb_ref")
    return p if os.path.exists(p) else None


def fold_hash(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return int(lib().orc_fold_hash_f32(_p(a), a.size))
    a = np.ascontiguousarray(a, dtype=np.int32)
    return int(lib().orc_fold_hash(_p(a), a.size))


# ----------------------------------------------------------------------------
# binary QC-LDPC
# ----------------------------------------------------------------------------
class BinaryCode:
    """Get_H + Transform_H (Simulation.cu:292-387) through the oracle."""

    def __init__(self, path, J, L, Z, literal=False):
        self.J, self.L, self.Z = J, L, Z
        self.N, self.M = L * Z, J * Z
        self.K = self.N - self.M
        self.H = np.zeros(J * L, np.int32)
        self.wc = np.zeros(J + 1, np.int32)
        self.wv = np.zeros(L + 1, np.int32)
        r = lib().orc_bldpc_get_h(path.encode(), J, L, _p(self.H), _p(self.wc), _p(self.wv))
        if r != 0:
            raise IOError("orc_bldpc_get_h(%s) -> %d" % (path, r))
        self.Wc, self.Wv = int(self.wc[J]), int(self.wv[L])
        self.addr = np.zeros(self.N * self.Wv, np.int32)
        lib().orc_bldpc_transform_h(_p(self.H), J, L, Z, _p(self.wc), _p(self.wv), _p(self.addr), 1 if literal else 0)


def bldpc_sigma(snr, snrtype=1, rate=0.0):
    return float(lib().orc_bldpc_sigma(np.float32(snr), snrtype, np.float32(rate)))


def bldpc_awgn(seed, sigma, N, F, codeword=None):
    """AWGNChannel_CPU (LDPC_Encoder.cu:25-43). seed: int32[3], advanced in place."""
    y = np.zeros(N * F, np.float32)
    cw = None if codeword is None else np.ascontiguousarray(codeword, np.int32)
    lib().orc_bldpc_awgn(_p(seed), c_float(sigma), _p(y), _p(cw), N, F)
    return y


def bldpc_decode(code, y, F, max_iter=50, early_exit=1, length=None, want_app=False, want_rq=False):
    """orc_bldpc_decode. Returns dict(D [(N+1)*F] int32, it, app, rq, flag_hist)."""
    y = np.ascontiguousarray(y, np.float32)
    assert y.size == code.N * F
    D = np.zeros((code.N + 1) * F, np.int32)
    app = np.zeros(code.N * F, np.float32) if want_app else None
    rq = np.zeros(F * code.M * code.Wc, np.float32) if want_rq else None
    hist = np.zeros(F, np.uint64)
    it = lib().orc_bldpc_decode(code.J, code.L, code.Z, _p(code.wc), _p(code.wv), _p(code.addr), _p(y), F, max_iter,
                                code.K if length is None else length, early_exit, _p(D), _p(app), _p(rq), _p(hist))
    return dict(D=D, it=int(it), app=app, rq=rq, flag_hist=hist)


def bldpc_statistic(counters, num_frames, D, N, F, length, itera_time, codeword=None, least_err=50, least_frames=10000):
    cw = None if codeword is None else np.ascontiguousarray(codeword, np.int32)
    return int(lib().orc_bldpc_statistic(_p(counters), num_frames, _p(cw), _p(D), N, F, length, itera_time, least_err, least_frames))


# ----------------------------------------------------------------------------
# non-binary EMS
# ----------------------------------------------------------------------------
class NBCode:
    """NB Get_H (Simulation.cpp:347-467) + GFInitial (GF.cpp:68-117) through the oracle."""

    def __init__(self, matrix_path, gf_path):
        dims = np.zeros(5, np.int32)
        r = lib().orc_nb_get_h(matrix_path.encode(), _p(dims), None, None, None, None, None, None)
        if r != 0:
            raise IOError("orc_nb_get_h(%s) -> %d" % (matrix_path, r))
        self.N, self.M, self.q, self.dv, self.dc = (int(x) for x in dims)
        self.m = int(np.log2(self.q))
        self.vn_w = np.zeros(self.N, np.int32)
        self.vn_cn = np.zeros(self.N * self.dv, np.int32)
        self.vn_gf = np.zeros(self.N * self.dv, np.int32)
        self.cn_w = np.zeros(self.M, np.int32)
        self.cn_vn = np.zeros(self.M * self.dc, np.int32)
        self.cn_gf = np.zeros(self.M * self.dc, np.int32)
        r = lib().orc_nb_get_h(matrix_path.encode(), _p(dims), _p(self.vn_w), _p(self.vn_cn), _p(self.vn_gf),
                               _p(self.cn_w), _p(self.cn_vn), _p(self.cn_gf))
        if r != 0:
            raise IOError("orc_nb_get_h(%s) -> %d" % (matrix_path, r))
        q = self.q
        self.mul = np.zeros(q * q, np.uint32)
        self.add = np.zeros(q * q, np.uint32)
        self.inv = np.zeros(q, np.uint32)
        r = lib().orc_gf_load(gf_path.encode(), q, _p(self.mul), _p(self.add), _p(self.inv))
        if r != 0:
            raise IOError("orc_gf_load(%s) -> %d" % (gf_path, r))
        self.rate = np.float32(self.N - self.M) / np.float32(self.N)


def nb_sigma(snr, rate, snrtype=0, n_qam=2):
    return float(lib().orc_nb_sigma(np.float32(snr), snrtype, n_qam, np.float32(rate)))


def nb_channel(code, cw_sym, seed, sigma):
    """Modulate + AWGNChannel_CPU + Demodulate for one frame -> (rx [N*m], Lch [N][q-1])."""
    cw = np.ascontiguousarray(cw_sym, np.int32)
    tx = np.zeros(code.N * code.m, np.float32)
    lib().orc_nb_modulate_bpsk(_p(cw), code.N, code.m, _p(tx))
    rx = np.zeros(code.N * code.m, np.float32)
    lib().orc_nb_awgn(_p(seed), c_float(sigma), _p(tx), _p(rx), code.N * code.m)
    return rx, nb_demodulate(code, rx, sigma)


def nb_read_constellation(path, n_points):
    """Get_CONSTELLATION -> float32 [n_points, 2] (Real, Image)."""
    con = np.zeros((n_points, 2), np.float32)
    r = lib().orc_nb_read_constellation(str(path).encode(), n_points, _p(con))
    if r != 0:
        raise IOError("orc_nb_read_constellation(%s) -> %d" % (path, r))
    return con


def nb_channel_qam(code, cw_sym, seed, sigma, con):
    """QAM branch (parity unpinned, see nbldpc_oracle.c): Modulate + AWGNChannel_CPU + Demodulate -> (rx [N,2], Lch [N][q-1])."""
    cw = np.ascontiguousarray(cw_sym, np.int32)
    con = np.ascontiguousarray(con, np.float32)
    rx = np.zeros((code.N, 2), np.float32)
    lib().orc_nb_awgn_qam(_p(seed), c_float(sigma), _p(cw), code.N, _p(con), _p(rx))
    return rx, nb_demodulate_qam(code, rx, sigma, con)


def nb_demodulate_qam(code, rx, sigma, con):
    rx = np.ascontiguousarray(rx, np.float32)
    con = np.ascontiguousarray(con, np.float32)
    Lch = np.zeros((code.N, code.q - 1), np.float32)
    lib().orc_nb_demodulate_qam(_p(rx), _p(con), c_float(sigma), code.N, code.q, _p(Lch))
    return Lch


def nb_demodulate(code, rx, sigma):
    rx = np.ascontiguousarray(rx, np.float32)
    Lch = np.zeros((code.N, code.q - 1), np.float32)
    lib().orc_nb_demodulate_bpsk(_p(rx), c_float(sigma), code.N, code.q, code.m, _p(Lch))
    return Lch


def nb_ems_decode(code, Lch, Nm=2, Nc=2, max_iter=20, dcmax_cfg=None, want_state=False):
    Lch = np.ascontiguousarray(Lch, np.float32)
    out = np.zeros(code.N, np.int32)
    it = c_int(0)
    LLR = np.zeros((code.N, code.q - 1), np.float32) if want_state else None
    c2v = np.zeros((code.M, code.dc, code.q - 1), np.float32) if want_state else None
    ok = lib().orc_nb_ems_decode(code.N, code.M, code.q, code.dv, code.dc, _p(code.vn_w), _p(code.vn_cn), _p(code.cn_w),
                                 _p(code.cn_vn), _p(code.cn_gf), _p(code.mul), _p(Lch), Nm, Nc, max_iter,
                                 code.dc if dcmax_cfg is None else dcmax_cfg, _p(out), ctypes.byref(it), _p(LLR), _p(c2v))
    return dict(out=out, it=it.value, ok=int(ok), LLR=LLR, c2v=c2v)


def nb_tmm_decode(code, Lch, max_iter=20, layered=False, want_state=False):
    """Decoding_TMM / Decoding_layered_TMM (decoder_method 1 / 3).  State: LLR [N][q], c2v [M][dc][q]."""
    Lch = np.ascontiguousarray(Lch, np.float32)
    out = np.zeros(code.N, np.int32)
    it = c_int(0)
    LLR = np.zeros((code.N, code.q), np.float32) if want_state else None
    c2v = np.zeros((code.M, code.dc, code.q), np.float32) if want_state else None
    ok = lib().orc_nb_tmm_decode(code.N, code.M, code.q, code.dv, code.dc, _p(code.vn_w), _p(code.vn_cn), _p(code.cn_w), _p(code.cn_vn),
                                 _p(code.cn_gf), _p(code.mul), _p(code.inv), _p(Lch), 1 if layered else 0, max_iter, _p(out),
                                 ctypes.byref(it), _p(LLR), _p(c2v))
    return dict(out=out, it=it.value, ok=int(ok), LLR=LLR, c2v=c2v)


def nb_ems_decode_batch(code, Lch, Nm=2, Nc=2, max_iter=20, dcmax_cfg=None):
    Lch = np.ascontiguousarray(Lch, np.float32)
    B = Lch.shape[0]
    out = np.zeros((B, code.N), np.int32)
    iters = np.zeros(B, np.int32)
    ok = np.zeros(B, np.int32)
    lib().orc_nb_ems_decode_batch(code.N, code.M, code.q, code.dv, code.dc, _p(code.vn_w), _p(code.vn_cn), _p(code.cn_w),
                                  _p(code.cn_vn), _p(code.cn_gf), _p(code.mul), _p(Lch), B, Nm, Nc, max_iter,
                                  code.dc if dcmax_cfg is None else dcmax_cfg, _p(out), _p(iters), _p(ok))
    return dict(out=out, it=iters, ok=ok)
