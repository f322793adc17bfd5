"""GPU parity tests of the binary QC-LDPC decoder (through the C ABI, include/bldpc.h).

Every comparison is bit-exact: hard bits, flags, iteration counts AND the float
a-posteriori sums (compared as uint32 bit patterns) against the CPU oracle
(oracle/bldpc_oracle.c, pinned in test_oracle_pins.py) and against the committed
golden fixtures.
"""
import os

import numpy as np
import pytest
import torch

from conftest import DATA, GOLDEN

pytestmark = pytest.mark.gpu

BL = os.path.join(DATA, "bldpc")


@pytest.fixture(scope="module")
def C():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cuda_ldpc_amd
    return cuda_ldpc_amd


def _path(J, L, Z):
    return os.path.join(BL, "J%d_L%d_Z%d_BlockH.txt" % (J, L, Z))


def _channel(orc, N, F, snr, seed=(173, 173, 173)):
    s = np.array(seed, np.int32)
    return orc.bldpc_awgn(s, orc.bldpc_sigma(snr), N, F)


def _decode(C, code, y, F, **kw):
    yt = torch.from_numpy(np.ascontiguousarray(y).reshape(code.N, F)).cuda()
    r = C.LDPC_Decoder_GPU(code, yt, **kw)
    torch.cuda.synchronize()
    out = dict(D=r["D"].cpu().numpy().reshape(-1), it=r["iteraTime"])
    out["app"] = None if r["app"] is None else r["app"].cpu().numpy().reshape(-1)
    out["flag_hist"] = None if r["flag_hist"] is None else r["flag_hist"].cpu().numpy().view(np.uint64)
    return out


def _assert_same(got, want, N, F, check_app=True):
    assert got["it"] == want["it"]
    assert np.array_equal(got["D"][: N * F], want["D"][: N * F]), "hard bits differ"
    assert np.array_equal(got["D"][N * F:], want["D"][N * F:]), "flag row differs"
    if check_app and got["app"] is not None:
        assert np.array_equal(got["app"].view(np.uint32), want["app"].view(np.uint32)), "a-posteriori LLRs differ bitwise"


KERNELS = ["table", "qc"]


def _k(C, name):
    return C.KERNEL_TABLE if name == "table" else C.KERNEL_QC_LDS


@pytest.mark.parametrize("kern", KERNELS)
@pytest.mark.parametrize("fn", ["bldpc_J4_L24_Z96_3dB_cor.npz", "bldpc_J32_L64_Z64_-1dB_cor.npz"])
def test_golden_fixture_batch_global(C, orc, kern, fn):
    """Reference semantics end to end: batch-global early exit, D + flags + iteraTime vs the committed fixture."""
    g = np.load(os.path.join(GOLDEN, fn))
    J, L, Z, F = int(g["J"]), int(g["L"]), int(g["Z"]), int(g["F"])
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    r = _decode(C, code, g["y"], F, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL, kernel=_k(C, kern))
    D = np.unpackbits(g["D_bits"])[: code.N * F].astype(np.int32)
    assert r["it"] == int(g["it"])
    assert np.array_equal(r["D"][: code.N * F], D)
    assert np.array_equal(r["D"][code.N * F:], g["flags"])
    assert orc.fold_hash(r["D"][: code.N * F]) == int(g["hash"])  # the hash SURVEY.md 8c recorded
    assert (kern == "qc") == code.last_kernel.startswith("qc_lds")


def test_golden_as_written_table(C, orc):
    """The reference's Transform_H as written (colliding slots): table kernel, level-scheduled VN order."""
    g = np.load(os.path.join(GOLDEN, "bldpc_J4_L24_Z96_4dB_lit.npz"))
    J, L, Z, F = 4, 24, 96, int(g["F"])
    H, wc, wv = C.Get_H(_path(J, L, Z), J, L)
    addr = C.Transform_H(H, J, L, Z, wc, wv, as_written=True)
    code = C.BinaryCode.from_table(J, L, Z, wc, wv, addr)
    assert code.levels > 1 and code.frames_per_wg == 0
    r = _decode(C, code, g["y"], F, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL)
    assert orc.fold_hash(r["D"][: code.N * F]) == 0x90C5DF9B
    assert np.array_equal(r["D"][code.N * F:], g["flags"]) and r["it"] == int(g["it"])
    with pytest.raises(Exception):
        _decode(C, code, g["y"], F, kernel=C.KERNEL_QC_LDS)


@pytest.mark.parametrize("kern", KERNELS)
@pytest.mark.parametrize("iters", [1, 2, 3, 7, 50])
def test_per_iteration_llr_bit_exact(C, orc, kern, iters):
    """Per-iteration a-posteriori LLRs (north_star: within 1e-5; we require bitwise equality)."""
    J, L, Z, F = 4, 24, 96, 16
    y = _channel(orc, L * Z, F, 3.0)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, iters, early_exit=0, want_app=True)
    got = _decode(C, code, y, F, max_iter=iters, exit_mode=C.EXIT_FIXED, kernel=_k(C, kern), want_app=True, want_flag_hist=True)
    _assert_same(got, want, code.N, F)
    mask = np.uint64((1 << min(iters, 64)) - 1)
    assert np.array_equal(got["flag_hist"] & mask, want["flag_hist"] & mask)


@pytest.mark.parametrize("kern", KERNELS)
@pytest.mark.parametrize("F", [1, 3, 4, 6, 37])
def test_ragged_batches(C, orc, kern, F):
    J, L, Z = 4, 24, 96
    y = _channel(orc, L * Z, F, 3.5)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    for mode, ee in ((C.EXIT_FIXED, 0), (C.EXIT_BATCH_GLOBAL, 1)):
        want = orc.bldpc_decode(ocode, y, F, 20, early_exit=ee, want_app=True)
        got = _decode(C, code, y, F, max_iter=20, exit_mode=mode, kernel=_k(C, kern), want_app=True)
        _assert_same(got, want, code.N, F)


# every matrix family the reference ships; kernel the product picks by itself
MATRICES = [(4, 24, 96, 3.0), (6, 24, 96, 2.0), (8, 24, 96, 1.0), (12, 24, 96, 0.0), (32, 64, 64, -0.5), (4, 24, 256, 3.0),
            (4, 24, 512, 3.0), (10, 60, 160, 2.5), (48, 60, 160, -3.0), (15, 30, 1280, -0.5)]


@pytest.mark.parametrize("J,L,Z,snr", MATRICES)
def test_matrix_families_auto_kernel(C, orc, J, L, Z, snr):
    F = 8
    y = _channel(orc, L * Z, F, snr)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 12, early_exit=0, want_app=True)
    got = _decode(C, code, y, F, max_iter=12, exit_mode=C.EXIT_FIXED, want_app=True)
    _assert_same(got, want, code.N, F)
    if code.frames_per_wg:  # also cross-check the generic kernel on the same code
        got2 = _decode(C, code, y, F, max_iter=12, exit_mode=C.EXIT_FIXED, kernel=C.KERNEL_TABLE, want_app=True)
        _assert_same(got2, want, code.N, F)


def test_pon_matrix(C, orc):
    J, L, Z, F = 12, 69, 256, 4
    p = os.path.join(BL, "PON_LDPC.txt")
    y = _channel(orc, L * Z, F, 2.0)
    ocode = orc.BinaryCode(p, J, L, Z)
    code = C.BinaryCode.from_blockh(p, J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 10, early_exit=0, want_app=True)
    got = _decode(C, code, y, F, max_iter=10, exit_mode=C.EXIT_FIXED, want_app=True)
    _assert_same(got, want, code.N, F)


# the fused kernels of the second and third tier (compressed check states in LDS / in registers): one code per variant
TIER_CODES = [("PON_LDPC.txt", 12, 69, 256, 2.6), ("J10_L60_Z160_BlockH.txt", 10, 60, 160, 3.2), ("J4_L24_Z512_BlockH.txt", 4, 24, 512, 3.6),
              ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, 0.4)]


@pytest.mark.parametrize("fn,J,L,Z,snr", TIER_CODES)
def test_tier_kernels_batch_global_and_history(C, orc, fn, J, L, Z, snr):
    """Reference early-exit rule (LDPC_Decoder.cu:150-153) and the per-iteration flag history on the compressed-state
    and register-state kernels, ragged batch."""
    F = 5
    p = os.path.join(BL, fn)
    y = _channel(orc, L * Z, F, snr)
    ocode = orc.BinaryCode(p, J, L, Z)
    code = C.BinaryCode.from_blockh(p, J, L, Z)
    assert code.frames_per_wg == 1
    want = orc.bldpc_decode(ocode, y, F, 40, early_exit=1, want_app=True)
    got = _decode(C, code, y, F, max_iter=40, exit_mode=C.EXIT_BATCH_GLOBAL, want_app=True, want_flag_hist=True)
    assert "compressed" in code.last_kernel or "regstate" in code.last_kernel
    _assert_same(got, want, code.N, F)
    mask = np.uint64((1 << want["it"]) - 1)
    assert np.array_equal(got["flag_hist"] & mask, want["flag_hist"] & mask)
    assert 1 < want["it"] < 40, "pick an SNR at which the batch converges before maxIT (it=%d)" % want["it"]
    # fixed iterations with the history on: every flag of every iteration
    want = orc.bldpc_decode(ocode, y, F, 9, early_exit=0, want_app=True)
    got = _decode(C, code, y, F, max_iter=9, exit_mode=C.EXIT_FIXED, want_app=True, want_flag_hist=True)
    _assert_same(got, want, code.N, F)
    assert np.array_equal(got["flag_hist"] & np.uint64(511), want["flag_hist"] & np.uint64(511))
    for it in (1, 2):
        want = orc.bldpc_decode(ocode, y, F, it, early_exit=0, want_app=True)
        _assert_same(_decode(C, code, y, F, max_iter=it, exit_mode=C.EXIT_FIXED, want_app=True), want, code.N, F)


@pytest.mark.parametrize("fn,J,L,Z,snr", TIER_CODES)
def test_tier_kernels_special_values(C, orc, fn, J, L, Z, snr):
    """Zeros, denormals, huge magnitudes and exact ties: the compressed states keep (min1, min2, index of the FIRST
    minimum, signs) -- duplicated minima and zero magnitudes are where that could differ from the per-edge messages."""
    F = 3
    rng = np.random.default_rng(11)
    N = L * Z
    y = rng.standard_normal(N * F).astype(np.float32)
    for val, cnt in ((0.0, N // 8), (-0.0, N // 8), (1e-41, N // 16), (-3e-42, N // 16), (3e38, N // 64), (-3e38, N // 64), (0.5, N // 2),
                     (-0.5, N // 2)):
        y[rng.integers(0, N * F, cnt)] = val
    p = os.path.join(BL, fn)
    ocode = orc.BinaryCode(p, J, L, Z)
    code = C.BinaryCode.from_blockh(p, J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 6, early_exit=0, want_app=True)
    got = _decode(C, code, y, F, max_iter=6, exit_mode=C.EXIT_FIXED, want_app=True)
    assert "compressed" in code.last_kernel or "regstate" in code.last_kernel
    _assert_same(got, want, code.N, F)
    got = _decode(C, code, y, F, max_iter=6, exit_mode=C.EXIT_FIXED, kernel=C.KERNEL_TABLE, want_app=True)
    _assert_same(got, want, code.N, F)


def test_special_values(C, orc):
    """Zeros, denormals, huge magnitudes, exact ties (duplicate minima) -- still bitwise equal."""
    J, L, Z, F = 4, 24, 96, 8
    rng = np.random.default_rng(7)
    N = L * Z
    y = rng.standard_normal(N * F).astype(np.float32)
    y[rng.integers(0, N * F, 2000)] = 0.0
    y[rng.integers(0, N * F, 2000)] = -0.0
    y[rng.integers(0, N * F, 2000)] = 1e-41       # denormal
    y[rng.integers(0, N * F, 2000)] = -3e-42
    y[rng.integers(0, N * F, 500)] = 3e38
    y[rng.integers(0, N * F, 500)] = -3e38
    y[rng.integers(0, N * F, 4000)] = 0.5          # ties
    y[rng.integers(0, N * F, 4000)] = -0.5
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 6, early_exit=0, want_app=True)
    for kern in KERNELS:
        got = _decode(C, code, y, F, max_iter=6, exit_mode=C.EXIT_FIXED, kernel=_k(C, kern), want_app=True)
        _assert_same(got, want, code.N, F)


def test_full_size_batch_properties(C, orc):
    """BASELINE config 2 at full size (65536 frames): a 64-frame oracle-checked block tiled 1024 times.

    Size-independent properties: every tile decodes to the same bits (frames are independent), and the
    first tile equals the oracle."""
    J, L, Z, Fb, reps = 4, 24, 96, 64, 1024
    N = L * Z
    y = _channel(orc, N, Fb, 3.0).reshape(N, Fb)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    want = orc.bldpc_decode(ocode, y.reshape(-1), Fb, 50, early_exit=0)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    yt = torch.from_numpy(y).cuda().repeat(1, reps).contiguous()
    r = C.LDPC_Decoder_GPU(code, yt, max_iter=50, exit_mode=C.EXIT_FIXED)
    torch.cuda.synchronize()
    assert code.last_kernel.startswith("qc_lds")
    D = r["D"].view(N + 1, reps, Fb)
    assert bool((D == D[:, :1, :]).all())
    got = D[:, 0, :].contiguous().cpu().numpy().reshape(-1)
    assert np.array_equal(got, want["D"])
    # all-zero codeword, noiseless: decodes to zero with every flag set
    clean = torch.ones((N, 256), dtype=torch.float32, device="cuda")
    r2 = C.LDPC_Decoder_GPU(code, clean, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL)
    assert r2["iteraTime"] == 1 and int(r2["D"][:N].sum()) == 0 and int(r2["D"][N].sum()) == 256


@pytest.mark.parametrize("J,L,Z,snr,Fb,reps", [(32, 64, 64, 0.0, 16, 2048), (15, 30, 1280, 0.0, 4, 2048)])
def test_full_size_batch_properties_configs_3_and_4(C, orc, J, L, Z, snr, Fb, reps):
    """BASELINE configs 3 (32 768 frames) and 4 (8 192 frames) at full size, 50 iterations: an oracle-checked block tiled over
    the batch decodes to the same columns in every tile, and the first tile equals the oracle bit for bit."""
    N = L * Z
    y = _channel(orc, N, Fb, snr).reshape(N, Fb)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    want = orc.bldpc_decode(ocode, y.reshape(-1), Fb, 50, early_exit=0)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    yt = torch.from_numpy(y).cuda().repeat(1, reps).contiguous()
    r = C.LDPC_Decoder_GPU(code, yt, max_iter=50, exit_mode=C.EXIT_FIXED)
    torch.cuda.synchronize()
    assert code.last_kernel.startswith("qc_lds")
    D = r["D"].view(N + 1, reps, Fb)
    assert bool((D == D[:, :1, :]).all())
    assert np.array_equal(D[:, 0, :].contiguous().cpu().numpy().reshape(-1), want["D"])
    del r, D, yt
    torch.cuda.empty_cache()


def test_statistic_matches_oracle(C, orc):
    J, L, Z, F = 4, 24, 96, 32
    y = _channel(orc, L * Z, F, 2.5)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 50, early_exit=1)
    cnt = np.zeros(5, np.int64)
    stop_want = orc.bldpc_statistic(cnt, F, want["D"], ocode.N, F, ocode.K, want["it"], least_err=5, least_frames=32)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    yt = torch.from_numpy(y.reshape(code.N, F)).cuda()
    r = C.LDPC_Decoder_GPU(code, yt, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL)
    SIM = C.SimCounters()
    SIM.num_Frames += F
    stop = C.Statistic(SIM, code, r["D"], r["iteraTime"], leastErrorFrames=5, leastTestFrames=32)
    assert [SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration, SIM.num_False_Frames, SIM.num_Alarm_Frames] == list(cnt)
    assert stop == stop_want


def test_argument_errors(C):
    J, L, Z = 4, 24, 96
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    y = torch.zeros((code.N, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(Exception):
        C.LDPC_Decoder_GPU(code, y, max_iter=0)
    with pytest.raises(Exception):
        C.LDPC_Decoder_GPU(code, y, length=code.N + 1)
    with pytest.raises(Exception):
        C.Get_H("/nonexistent/file.txt", J, L)
    bad = np.full(J * L, Z + 5, np.int32)
    with pytest.raises(Exception):
        C.BinaryCode.from_shifts(bad, J, L, Z)


def test_simulation_gpu_loop_matches_oracle(C, orc):
    """Simulation_GPU (Simulation.cu:12-171): batch loop + stop rule, counters vs a CPU replay of the same loop."""
    from cuda_ldpc_amd.simulation import Simulation_GPU, format_row
    J, L, Z, F = 4, 24, 96, 64
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    snr = 2.8
    seed = np.array([173, 173, 173], np.int32)
    SIM = C.SimCounters()
    SIM.SNR = snr
    rows = []
    stop = Simulation_GPU(code, seed, C.sigma_of(snr), SIM, Num_Frames_OneTime=F, maxIT=50, leastErrorFrames=3, leastTestFrames=128,
                          displayStep=F, log=rows.append)
    assert stop == 1
    # CPU replay
    oseed = np.array([173, 173, 173], np.int32)
    cnt = np.zeros(5, np.int64)
    frames = 0
    while True:
        frames += F
        y = orc.bldpc_awgn(oseed, orc.bldpc_sigma(snr), ocode.N, F)
        r = orc.bldpc_decode(ocode, y, F, 50, early_exit=1)
        if orc.bldpc_statistic(cnt, frames, r["D"], ocode.N, F, ocode.K, r["it"], least_err=3, least_frames=128):
            break
    assert frames == SIM.num_Frames and np.array_equal(seed, oseed)
    assert [SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration, SIM.num_False_Frames, SIM.num_Alarm_Frames] == cnt.tolist()
    assert len(rows) == frames // F and rows[-1] == format_row(SIM, code.K)


def test_device_channel_same_draws_as_host(C, orc):
    """bldpc_awgn_channel_device: LCG jump-ahead reproduces the serial stream; only the device libm may move a
    sample by an ulp.  Seeds after the batch are identical; decoding both inputs gives the same error statistics."""
    N, F = 2304, 64
    sigma = C.sigma_of(3.0)
    s_host = np.array([173, 173, 173], np.int32)
    s_dev = s_host.copy()
    yh = C.AWGNChannel_CPU(s_host, sigma, N, F)
    yd = C.AWGNChannel_GPU(s_dev, sigma, N, F).cpu().numpy()
    assert np.array_equal(s_host, s_dev)
    same = (yh.view(np.int32) == yd.view(np.int32)).mean()
    assert same > 0.9 and np.abs(yh - yd).max() < 1e-6, (same, np.abs(yh - yd).max())  # measured: 91.5 % bit-identical
    # a second batch continues the same stream
    yh2 = C.AWGNChannel_CPU(s_host, sigma, N, 8)
    yd2 = C.AWGNChannel_GPU(s_dev, sigma, N, 8).cpu().numpy()
    assert np.abs(yh2 - yd2).max() < 1e-6 and np.array_equal(s_host, s_dev)
    # with a transmitted codeword
    cw = np.random.default_rng(1).integers(0, 2, (N, 8)).astype(np.int32)
    s1 = np.array([5, 6, 7], np.int32); s2 = s1.copy()
    a = C.AWGNChannel_CPU(s1, 0.5, N, 8, CodeWord=cw)
    b = C.AWGNChannel_GPU(s2, 0.5, N, 8, CodeWord=torch.from_numpy(cw).cuda()).cpu().numpy()
    assert np.abs(a - b).max() < 1e-6


def test_reference_signature_shim_cpp_harness(C, tmp_path):
    """A C++ harness in the reference's calling style drives LDPC_Decoder_GPU (reference signature, shim/) end to end:
    the hashes SURVEY 8c recorded from the reference kernels come out (corrected table and table as written)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from cuda_ldpc_amd._lib import SO_PATH
    root = os.path.dirname(os.path.dirname(SO_PATH))
    exe = str(tmp_path / "ref_style_harness")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "shim"),
                           os.path.join(root, "tests", "cpp", "ref_style_harness.cpp"), os.path.join(root, "shim", "ldpc_ref_shim.hip"),
                           "-o", exe, "-L", os.path.dirname(SO_PATH), "-lcuda_ldpc_amd", "-Wl,-rpath," + os.path.dirname(SO_PATH)])
    m = _path(4, 24, 96)
    out = subprocess.check_output([exe, m, "4", "24", "96", "32", "3.0", "0"]).decode()
    assert "hash=05a41534 iteraTime=50 flags=31/32" in out, out
    out = subprocess.check_output([exe, m, "4", "24", "96", "32", "4.0", "0"]).decode()
    assert "hash=99f71dc5 iteraTime=6 flags=32/32" in out, out
    out = subprocess.check_output([exe, m, "4", "24", "96", "32", "4.0", "1"]).decode()
    assert "hash=90c5df9b iteraTime=50" in out, out


def _build_cpp(root, so_path, tmp_path, name, sources):
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / name)
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "shim")]
                          + [os.path.join(root, s) for s in sources] + ["-o", exe, "-L", os.path.dirname(so_path), "-lcuda_ldpc_amd",
                                                                         "-Wl,-rpath," + os.path.dirname(so_path)])
    return exe


@pytest.fixture(scope="module")
def bmain_exe(C, tmp_path_factory):
    """tests/cpp/ref_main_style_sweep.cpp + shim/ldpc_ref_shim.hip linked against the library: built once per module."""
    from cuda_ldpc_amd._lib import SO_PATH
    root = os.path.dirname(os.path.dirname(SO_PATH))
    return _build_cpp(root, SO_PATH, tmp_path_factory.mktemp("bmain"), "ref_main_style_sweep", ["tests/cpp/ref_main_style_sweep.cpp", "shim/ldpc_ref_shim.hip"])


@pytest.mark.parametrize("as_written", [0, 1])
def test_reference_main_style_sweep_cpp_harness(C, orc, bmain_exe, as_written):
    """A sweep written like the reference's main() (main.cu:114-160) drives Get_H, Transform_H, Simulation_GPU, Statistic and
    LDPC_Decoder_GPU with the reference's signatures and structs (shim/ldpc_ref_shim.hpp).  Every SNR point's counters equal
    a CPU replay of the same loop through the oracle; with the intended circulant table the shim reaches the fused kernel
    (it reads the shifts back from the table), with the table as written the table kernels."""
    import subprocess
    from cuda_ldpc_amd._lib import SO_PATH
    root = os.path.dirname(os.path.dirname(SO_PATH))
    exe = bmain_exe
    J, L, Z, F, maxIT = 4, 24, 96, 256, 50
    out = subprocess.check_output([exe, _path(J, L, Z), str(J), str(L), str(Z), str(F), str(maxIT), str(as_written), "3.0", "3.5", "0.2",
                                   "3", "512"]).decode()
    pts = [ln.split() for ln in out.splitlines() if ln.startswith("POINT")]
    assert len(pts) == 3 and "task finish" in out  # 3.0, 3.2 (3.20000005), 3.4 (3.4000001): a float advanced by a double step
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z, literal=bool(as_written))
    snr = np.float32(3.0)
    for p in pts:
        assert abs(float(p[1]) - float(snr)) < 1e-6
        oseed = np.array([173, 173, 173], np.int32)
        cnt = np.zeros(5, np.int64)
        frames = 0
        while True:
            frames += F
            y = orc.bldpc_awgn(oseed, orc.bldpc_sigma(float(snr)), ocode.N, F)
            r = orc.bldpc_decode(ocode, y, F, maxIT, early_exit=1)
            if orc.bldpc_statistic(cnt, frames, r["D"], ocode.N, F, ocode.K, r["it"], least_err=3, least_frames=512):
                break
        assert [int(x) for x in p[2:8]] == [frames] + cnt.tolist(), (p, frames, cnt)
        assert p[8].startswith("kernel=table" if as_written else "kernel=qc_lds"), p[8]
        snr = np.float32(np.float64(snr) + 0.2)
    rows = [ln for ln in out.splitlines() if ln.startswith(" 3.")]
    assert len(rows) >= 3  # the reference's result rows (Simulation.cu:272)


@pytest.mark.parametrize("exit_mode", [2, 1, 0])
def test_reference_main_style_sweep_fast_path(C, bmain_exe, exit_mode):
    """The fast path through the binary drop-in boundary (bldpc_shim_configure_fast): the same main.cu-style C++ sweep with the
    device-side channel and bldpc_decode_statistic behind the reference-signature Simulation_GPU -- no host channel, no copy of D,
    no host loop.  Counters per point equal the Python mirror (cuda_ldpc_amd.simulation.Simulation_GPU) with the same switches:
    same draws, same device libm, same kernels."""
    import subprocess
    from cuda_ldpc_amd._lib import SO_PATH
    from cuda_ldpc_amd.simulation import Simulation_GPU
    root = os.path.dirname(os.path.dirname(SO_PATH))
    exe = bmain_exe
    J, L, Z, F, maxIT = 4, 24, 96, 4096, 50
    out = subprocess.check_output([exe, _path(J, L, Z), str(J), str(L), str(Z), str(F), str(maxIT), "0", "3.0", "3.3", "0.2", "20", "8192",
                                   "1", "1", str(exit_mode), "0"]).decode()
    pts = [ln.split() for ln in out.splitlines() if ln.startswith("POINT")]
    assert len(pts) == 2 and "task finish" in out
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    snr = np.float32(3.0)
    for p in pts:
        seed = np.array([173, 173, 173], np.int32)
        SIM = C.SimCounters()
        SIM.SNR = float(snr)
        Simulation_GPU(code, seed, C.sigma_of(float(snr)), SIM, Num_Frames_OneTime=F, maxIT=maxIT, exit_mode=exit_mode, leastErrorFrames=20,
                       leastTestFrames=8192, device_channel=True, log=None)
        assert [int(x) for x in p[2:8]] == [SIM.num_Frames, SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration, SIM.num_False_Frames,
                                            SIM.num_Alarm_Frames], (p, SIM)
        assert p[8].startswith("kernel=qc_lds"), p[8]
        snr = np.float32(np.float64(snr) + 0.2)


def test_reference_main_style_sweep_fast_path_reproduces_the_committed_sweep(C, bmain_exe):
    """Two Es/N0 points of profiles/r02c_sweep_binary_J4_L24_Z96_per_frame_exit.txt (sweep.py binary --device-channel --per-frame
    --batch 262144), count for count, from the C++ harness: 3.0 dB 262144 frames / 575 error frames, 3.2 dB 262144 / 56."""
    import subprocess
    from cuda_ldpc_amd._lib import SO_PATH
    root = os.path.dirname(os.path.dirname(SO_PATH))
    exe = bmain_exe
    out = subprocess.check_output([exe, _path(4, 24, 96), "4", "24", "96", "262144", "50", "0", "3.0", "3.3", "0.2", "50", "10000",
                                   "1", "1", "2", "1200"]).decode()
    rows = [ln.split() for ln in out.splitlines() if ln.startswith(" 3.")]
    assert [r[:3] for r in rows] == [["3.0", "262144", "575"], ["3.2", "262144", "56"]], out
    assert rows[0][3:6] == ["2.1935e-03", "3.5749e-05", "6.65"] and rows[1][3:6] == ["2.1362e-04", "3.0259e-06", "5.36"], rows


def test_max_iter_above_64_falls_back_to_the_table_kernels(C, orc):
    """The reference takes any maxIT (define.cuh:35).  The fused kernels keep 64 iterations of flag history, so
    KERNEL_AUTO sends a batch-global decode with max_iter > 64 to the table kernels instead of refusing it; an explicit
    QC_LDS request is still refused."""
    J, L, Z, F = 4, 24, 96, 16
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    y = _channel(orc, code.N, F, 2.0)  # a frame that never passes: the batch runs to max_iter
    want = orc.bldpc_decode(ocode, y, F, 70, early_exit=1, want_app=True)
    assert want["it"] == 70
    got = _decode(C, code, y, F, max_iter=70, exit_mode=C.EXIT_BATCH_GLOBAL, want_app=True)
    assert code.last_kernel.startswith("table")
    _assert_same(got, want, code.N, F)
    with pytest.raises(Exception):
        _decode(C, code, y, F, max_iter=70, exit_mode=C.EXIT_BATCH_GLOBAL, kernel=C.KERNEL_QC_LDS)
    got = _decode(C, code, y, F, max_iter=70, exit_mode=C.EXIT_FIXED, want_app=True)  # fixed iterations: no history needed
    assert code.last_kernel.startswith("qc_lds")
    _assert_same(got, orc.bldpc_decode(ocode, y, F, 70, early_exit=0, want_app=True), code.N, F)


def test_bench_two_ranks_share_one_gpu_over_gloo(C):
    """The N > 1 path of bench.py as the driver launches it (python -m torch.distributed.run, one process per rank), rehearsed on
    this box's single GPU: BENCH_SINGLE_DEVICE=1 puts both ranks on cuda:0, BENCH_DIST_BACKEND=gloo carries the one collective of
    the path (the all-reduce of the 5 error counters).  Weak scaling: every rank decodes its own 2048 frames of the same tiled
    block, so the job-wide counters are exactly twice the single-rank ones."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--frames", "2048", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.check_output([sys.executable, os.path.join(root, "bench.py")] + common, stderr=subprocess.DEVNULL).decode()
    j1 = json.loads([ln for ln in one.splitlines() if ln.startswith("{")][0])
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    two = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                   "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                                  stderr=subprocess.DEVNULL, env=env, timeout=600).decode()
    lines = [ln for ln in two.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints, once
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["value"] > 0
    for k in ("frames", "error_frames", "error_bits"):
        assert j2["stats"][k] == 2 * j1["stats"][k], (k, j1["stats"], j2["stats"])
    assert j1["stats"]["error_frames"] > 0


def test_rccl_world_size_one_smoke(C):
    """RCCL on the one GPU there is: ONE fresh child process (python -m torch.distributed.run --nproc-per-node 1) initialises
    init_process_group("nccl", device_id=cuda:0) before it touches the GPU, runs Simulation_GPU(..., dist=dist) for two batches and
    the counter all-reduce of sharding.allreduce_counters; its counters equal the un-distributed run made here.  Proves that
    librccl loads and the collective of the multi-GPU path executes on MI355X (not a scaling measurement)."""
    import json
    import socket
    import subprocess
    import sys
    from cuda_ldpc_amd.simulation import Simulation_GPU
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "rccl_world1_child.py")], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["backend"] == "nccl" and j["world"] == 1 and j["rccl_loaded"] and j["probe"] == [0, 1, 2, 3, 4]
    code = C.BinaryCode.from_blockh(_path(4, 24, 96), 4, 24, 96)
    seed = np.array([173, 173, 173], np.int32)
    SIM = C.SimCounters()
    Simulation_GPU(code, seed, C.sigma_of(3.0), SIM, Num_Frames_OneTime=2048, maxIT=50, exit_mode=C.EXIT_PER_FRAME, max_batches=2, log=None,
                   device_channel=True)
    assert [j["frames"], j["error_frames"], j["error_bits"], j["total_iteration"], j["seed"]] == \
        [SIM.num_Frames, SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration, seed.tolist()]
    assert j["frames"] == 4096 and j["error_frames"] > 0


def test_bench_contract_line(C):
    """bench.py prints ONE JSON line with the driver's contract fields plus `roofline` and `cpu_baseline`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--frames", "2048", "--steps", "2", "--warmup", "1",
                                   "--cpu-frames", "32"], stderr=subprocess.DEVNULL).decode().strip().splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["unit"] == "codewords/s" and j["value"] > 0 and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]


@pytest.mark.parametrize("Z,J,L", [(64, 5, 12), (96, 3, 9), (128, 6, 20), (192, 4, 16), (320, 5, 11), (384, 3, 24), (640, 4, 10), (1024, 3, 8)])
def test_other_lifting_sizes_random_matrices(C, orc, tmp_path, Z, J, L):
    """Random block matrices with lifting sizes the reference's matrix set does not contain: the generic compressed-state
    kernel is picked (these shapes have no ahead-of-time messages-in-LDS variant) and agrees with the oracle and the table kernels."""
    rng = np.random.default_rng(Z + J)
    H = rng.integers(0, Z, size=(J, L)).astype(np.int32)
    H[rng.random((J, L)) < 0.45] = -1
    for l in range(L):  # every column keeps at least two blocks, every row at least three
        rows = rng.permutation(J)[:2]
        for r in rows:
            if H[r, l] < 0:
                H[r, l] = rng.integers(0, Z)
    for j in range(J):
        cols = rng.permutation(L)[:3]
        for c in cols:
            if H[j, c] < 0:
                H[j, c] = rng.integers(0, Z)
    path = str(tmp_path / "H.txt")
    with open(path, "w") as f:
        for j in range(J):
            f.write("\t".join(str(int(x)) for x in H[j]) + "\r\n")
    F = 5
    y = _channel(orc, L * Z, F, 1.0)
    ocode = orc.BinaryCode(path, J, L, Z)
    code = C.BinaryCode.from_blockh(path, J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 8, early_exit=0, want_app=True)
    got = _decode(C, code, y, F, max_iter=8, exit_mode=C.EXIT_FIXED, want_app=True)
    assert "compressed" in code.last_kernel, code.last_kernel
    _assert_same(got, want, code.N, F)
    got = _decode(C, code, y, F, max_iter=8, exit_mode=C.EXIT_FIXED, kernel=C.KERNEL_TABLE, want_app=True)
    _assert_same(got, want, code.N, F)
    want = orc.bldpc_decode(ocode, y, F, 30, early_exit=1, want_app=True)
    got = _decode(C, code, y, F, max_iter=30, exit_mode=C.EXIT_BATCH_GLOBAL, want_app=True)
    _assert_same(got, want, code.N, F)
    Dw, appw, itw = _oracle_per_frame(orc, ocode, y, F, 30)  # and every frame on its own flag
    D, app, it, _ = _decode_per_frame(C, code, y, F, 30, C.KERNEL_QC_LDS)
    assert np.array_equal(it, itw) and np.array_equal(D, Dw) and np.array_equal(app.view(np.uint32), appw.view(np.uint32))


def _write_blockh(path, H):
    with open(path, "w") as f:
        for row in H:
            f.write("\t".join(str(int(x)) for x in row) + "\r\n")


LOCAL_CASES = [  # (shape of, J, L, Z, how the pattern is made, tag expected in the kernel name)
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, "shifts", "halfrow-local"),    # the reference's block pattern, other shifts
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, "pattern", "halfrow-local"),   # full rows, other zero blocks: another matching, light columns
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, "lightrow", "halfrow<"),       # a row of 19 blocks: no local edges, the plain half-row kernel
    ("J32_L64_Z64_BlockH.txt", 32, 64, 64, "shifts", "row-local"),
    ("J32_L64_Z64_BlockH.txt", 32, 64, 64, "permuted", "row-local"),    # block rows permuted: other row weights per thread group, other places of the local blocks
    ("J32_L64_Z64_BlockH.txt", 32, 64, 64, "lightcol", "row<"),         # a column of weight 2: the plain row kernel
]


@pytest.mark.parametrize("fn,J,L,Z,how,tag", LOCAL_CASES)
def test_local_edge_kernels_on_other_matrices(C, orc, tmp_path, monkeypatch, fn, J, L, Z, how, tag):
    """k_qc / k_qc2 with local edges (one block per column stays in the check thread's registers, the column stored rotated): on
    matrices other than the two the reference ships in these shapes -- other shifts, other block patterns (other matchings, other
    places of the local block in its column's order, columns lighter than WV) -- the outputs equal the oracle's bit for bit
    (hard bits, a-posteriori sums, flags; fixed, batch-global and per-frame exits) and those of the same code created under
    BLDPC_NO_LOCAL=1; patterns the local-edge form does not take (a light row / a light column) fall back to the plain kernels.  The
    batch-global rule's full runs (flag history on) use the local-edge kernels; the per-frame passes too on the half-row kernel, while
    the row kernel runs them on the plain plan the code object carries beside the local-edge one."""
    base = np.loadtxt(os.path.join(BL, fn), dtype=np.int64).reshape(J, L)
    rng = np.random.default_rng(J * 1000 + len(how))
    H = np.where(base >= 0, rng.integers(0, Z, size=(J, L)), -1)
    if how == "pattern":  # four zero blocks per row at random, every column keeps at least two blocks
        while True:
            H = rng.integers(0, Z, size=(J, L))
            for j in range(J):
                H[j, rng.permutation(L)[:4]] = -1
            if ((H >= 0).sum(0) >= 2).all() and ((H >= 0).sum(0) < J).any():
                break
    elif how == "lightrow":
        H[1, int(np.argmax(H[1] >= 0))] = -1
    elif how == "permuted":
        H = H[rng.permutation(J)]
    elif how == "lightcol":
        H[int(np.argmax(H[:, 5] >= 0)), 5] = -1
    path = str(tmp_path / "H.txt")
    _write_blockh(path, H)
    F = 5 if how in ("pattern", "permuted") else 6  # odd F: the regrouped input instead of the in-place read
    snr = 3.0 if J == 4 else 0.5
    y = _channel(orc, L * Z, F, snr)
    ocode = orc.BinaryCode(path, J, L, Z)
    code = C.BinaryCode.from_blockh(path, J, L, Z)
    monkeypatch.setenv("BLDPC_NO_LOCAL", "1")
    plain = C.BinaryCode.from_blockh(path, J, L, Z)
    monkeypatch.delenv("BLDPC_NO_LOCAL")
    for its in (1, 2, 7):
        want = orc.bldpc_decode(ocode, y, F, its, early_exit=0, want_app=True)
        got = _decode(C, code, y, F, max_iter=its, exit_mode=C.EXIT_FIXED, want_app=True)
        assert tag in code.last_kernel, code.last_kernel
        _assert_same(got, want, code.N, F)
        ref = _decode(C, plain, y, F, max_iter=its, exit_mode=C.EXIT_FIXED, want_app=True)
        assert "local" not in plain.last_kernel, plain.last_kernel
        _assert_same(ref, want, code.N, F)
    want = orc.bldpc_decode(ocode, y, F, 30, early_exit=1, want_app=True)
    got = _decode(C, code, y, F, max_iter=30, exit_mode=C.EXIT_BATCH_GLOBAL, want_app=True, want_flag_hist=True)
    _assert_same(got, want, code.N, F)
    Dw, appw, itw = _oracle_per_frame(orc, ocode, y, F, 30)
    D, app, it, _ = _decode_per_frame(C, code, y, F, 30, C.KERNEL_QC_LDS)
    assert np.array_equal(it, itw) and np.array_equal(D, Dw) and np.array_equal(app.view(np.uint32), appw.view(np.uint32))
    # the per-frame exit: the half-row kernel keeps its local-edge form (persistent for large batches), the row kernel runs the plain
    # plan the code object carries beside the local-edge one (QcPlan::pf)
    assert "qc_lds" in code.last_kernel and ("local" in code.last_kernel) == (tag == "halfrow-local")


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_long_block_kernels_on_random_shifts(C, orc, tmp_path, monkeypatch, seed):
    """J15_L30_Z1280's block pattern with RANDOM shifts: k_qcr2 (halo columns, tabulated wave-linear offsets, the wrapped blocks of a
    (row, tile) in the last two or three, per-lane, slots; seed 3 needs the three-slot instantiation) must meet every wrap position,
    and a matrix with more wrapped blocks in one (row, tile) must fall back to k_qcr; both kernels against the oracle (a-posteriori bits, hard decisions, flags, history and
    per-frame exits) and against each other on the same code."""
    J, L, Z = 15, 30, 1280
    base = np.loadtxt(os.path.join(BL, "J15_L30_Z1280_BlockH.txt"), dtype=np.int64).reshape(J, L)
    rng = np.random.default_rng(seed)
    H = np.where(base >= 0, rng.integers(0, Z, size=(J, L)), -1)
    if seed % 2 == 0:  # a row's shifts a few positions past the same tile boundary: all of its blocks wrap in the same tile
        H = np.where(base >= 0, (64 * rng.integers(0, Z // 64, size=(J, 1)) + rng.integers(1, 6, size=(J, L))) % Z, -1)
        lc = int(np.argmax((base >= 0).sum(0) == J))
        H[:, lc] = np.where(base[:, lc] >= 0, 0, -1)  # (the kernels see shifts relative to this column)
    path = str(tmp_path / "H.txt")
    with open(path, "w") as f:
        for j in range(J):
            f.write("\t".join(str(int(x)) for x in H[j]) + "\r\n")
    F = 3
    y = _channel(orc, L * Z, F, 0.2)
    ocode = orc.BinaryCode(path, J, L, Z)
    code = C.BinaryCode.from_blockh(path, J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 6, early_exit=0, want_app=True)
    got = _decode(C, code, y, F, max_iter=6, exit_mode=C.EXIT_FIXED, want_app=True)
    assert ("regstate-halo" in code.last_kernel) == (seed % 2 == 1), code.last_kernel
    assert "regstate" in code.last_kernel
    _assert_same(got, want, code.N, F)
    want = orc.bldpc_decode(ocode, y, F, 25, early_exit=1, want_app=True)
    got = _decode(C, code, y, F, max_iter=25, exit_mode=C.EXIT_BATCH_GLOBAL, want_app=True)
    _assert_same(got, want, code.N, F)
    Dw, appw, itw = _oracle_per_frame(orc, ocode, y, F, 25)
    D, app, it, _ = _decode_per_frame(C, code, y, F, 25, C.KERNEL_QC_LDS)
    assert np.array_equal(it, itw) and np.array_equal(D, Dw) and np.array_equal(app.view(np.uint32), appw.view(np.uint32))
    if seed % 2 == 1:  # the same code pinned to k_qcr
        monkeypatch.setenv("BLDPC_NO_HALO", "1")
        old = C.BinaryCode.from_blockh(path, J, L, Z)
        monkeypatch.delenv("BLDPC_NO_HALO")
        want = orc.bldpc_decode(ocode, y, F, 6, early_exit=0, want_app=True)
        got = _decode(C, old, y, F, max_iter=6, exit_mode=C.EXIT_FIXED, want_app=True)
        assert "regstate<" in old.last_kernel, old.last_kernel
        _assert_same(got, want, code.N, F)


@pytest.mark.parametrize("J,L,Z,snr,F", [(4, 24, 96, 2.2, 70), (4, 24, 96, 2.2, 37), (32, 64, 64, -0.8, 18), (15, 30, 1280, -0.4, 5)])
def test_decode_statistic_equals_the_two_calls(C, orc, J, L, Z, snr, F):
    """bldpc_decode_statistic = LDPC_Decoder_GPU + Statistic (Simulation.cu:143-145) in one call: D, iteration counts and the five
    counters must equal those of the two calls in every exit mode, on the fused kernels (errors counted from the packed hard bits:
    whole and partial words of `length`, even and odd batch sizes) and on the table kernels (plain sequence), accumulated over
    two batches."""
    import ctypes
    from cuda_ldpc_amd._lib import check, lib
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    y = torch.from_numpy(np.ascontiguousarray(_channel(orc, L * Z, F, snr)).reshape(L * Z, F)).cuda()
    for kern in (C.KERNEL_AUTO, C.KERNEL_TABLE):
        for mode in (C.EXIT_FIXED, C.EXIT_PER_FRAME, C.EXIT_BATCH_GLOBAL):
            for length in (0, 100, code.N, 1):
                a = torch.zeros(5, dtype=torch.int64, device="cuda")
                b = torch.zeros(5, dtype=torch.int64, device="cuda")
                for _ in range(2):
                    r = C.LDPC_Decoder_GPU(code, y, max_iter=9, length=length, exit_mode=mode, kernel=kern)
                    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                    if mode == C.EXIT_PER_FRAME:
                        check(lib.bldpc_statistic_per_frame(code._h, ctypes.c_void_p(r["D"].data_ptr()), None, F, length,
                                                            ctypes.c_void_p(r["iters"].data_ptr()), ctypes.c_void_p(a.data_ptr()), st), "Statistic")
                    else:
                        check(lib.bldpc_statistic(code._h, ctypes.c_void_p(r["D"].data_ptr()), None, F, length, r["iteraTime"],
                                                  ctypes.c_void_p(a.data_ptr()), st), "Statistic")
                    q = C.Decode_Statistic(code, y, b, max_iter=9, length=length, exit_mode=mode, kernel=kern)
                    torch.cuda.synchronize()
                    assert torch.equal(q["D"], r["D"]) and q["iteraTime"] == r["iteraTime"]
                    if mode == C.EXIT_PER_FRAME:
                        assert torch.equal(q["iters"], r["iters"])
                assert torch.equal(a, b), (kern, mode, length, a.tolist(), b.tolist())
                assert int(a[1]) > 0 or length in (1, 100)  # the batch does hold bit errors


# ---- per-frame termination (bldpc_decode_per_frame): the reference rule on batches of one frame -------------------------
def _oracle_per_frame(orc, ocode, y, F, max_iter):
    """LDPC_Decoder.cu:94-156 run on every frame alone (Num_Frames_OneTime = 1): D column, flag, iteraTime, sums per frame."""
    N = ocode.N
    yy = np.ascontiguousarray(y, np.float32).reshape(N, F)
    D = np.zeros((N + 1, F), np.int32)
    app = np.zeros((N, F), np.float32)
    iters = np.zeros(F, np.int32)
    for f in range(F):
        w = orc.bldpc_decode(ocode, np.ascontiguousarray(yy[:, f]), 1, max_iter, early_exit=1, want_app=True)
        D[:, f] = w["D"]
        app[:, f] = w["app"]
        iters[f] = w["it"]
    return D, app, iters


def _decode_per_frame(C, code, y, F, max_iter, kernel):
    yt = torch.from_numpy(np.ascontiguousarray(y).reshape(code.N, F)).cuda()
    r = C.LDPC_Decoder_GPU(code, yt, max_iter=max_iter, exit_mode=C.EXIT_PER_FRAME, kernel=kernel, want_app=True)
    torch.cuda.synchronize()
    return r["D"].cpu().numpy(), r["app"].cpu().numpy(), r["iters"].cpu().numpy(), r


PER_FRAME_CODES = [("J4_L24_Z96_BlockH.txt", 4, 24, 96, 2.6, 37, 30), ("J32_L64_Z64_BlockH.txt", 32, 64, 64, -0.9, 21, 30),
                   ("J4_L24_Z256_BlockH.txt", 4, 24, 256, 2.8, 9, 30), ("J10_L60_Z160_BlockH.txt", 10, 60, 160, 2.9, 5, 30),
                   ("PON_LDPC.txt", 12, 69, 256, 2.3, 5, 30), ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, 0.1, 3, 30)]


@pytest.mark.parametrize("fn,J,L,Z,snr,F,maxit", PER_FRAME_CODES)
def test_per_frame_exit_matches_reference_rule_on_single_frames(C, orc, fn, J, L, Z, snr, F, maxit):
    """Every frame stops on its own flag: hard bits, flag, a-posteriori sums and iteration count of each frame equal the
    oracle's decode of that frame alone under the reference's early-exit rule -- on every kernel tier, ragged batches, with
    frames that stop at different iterations inside one workgroup and frames that never stop."""
    p = os.path.join(BL, fn)
    y = _channel(orc, L * Z, F, snr)
    ocode = orc.BinaryCode(p, J, L, Z)
    code = C.BinaryCode.from_blockh(p, J, L, Z)
    Dw, appw, itw = _oracle_per_frame(orc, ocode, y, F, maxit)
    assert len(set(itw.tolist())) > 1, "pick an SNR at which frames stop at different iterations (%s)" % itw
    for kern in (C.KERNEL_QC_LDS, C.KERNEL_TABLE):
        D, app, it, _ = _decode_per_frame(C, code, y, F, maxit, kern)
        assert np.array_equal(it, itw), "%s: iteration counts differ %s vs %s" % (code.last_kernel, it, itw)
        assert np.array_equal(D, Dw), "%s: hard bits / flags differ" % code.last_kernel
        assert np.array_equal(app.view(np.uint32), appw.view(np.uint32)), "%s: a-posteriori sums differ" % code.last_kernel


def test_per_frame_exit_edge_cases(C, orc):
    """max_iter = 1 and 2, a batch in which no frame ever stops, one in which every frame stops at iteration 1."""
    J, L, Z = 4, 24, 96
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    for snr, F, maxit in ((2.6, 7, 1), (2.6, 7, 2), (-6.0, 6, 5), (12.0, 9, 8)):
        y = _channel(orc, L * Z, F, snr)
        Dw, appw, itw = _oracle_per_frame(orc, ocode, y, F, maxit)
        for kern in (C.KERNEL_QC_LDS, C.KERNEL_TABLE):
            D, app, it, _ = _decode_per_frame(C, code, y, F, maxit, kern)
            assert np.array_equal(it, itw) and np.array_equal(D, Dw) and np.array_equal(app.view(np.uint32), appw.view(np.uint32))
    assert set(itw.tolist()) == {1}  # the last case: clean channel, hard decision of the channel values already passes
    import ctypes
    from cuda_ldpc_amd._lib import LdpcError, check, lib
    with pytest.raises(LdpcError):  # the batch entry point has one iteration count: it refuses the per-frame mode
        yt = torch.zeros((code.N, 4), device="cuda")
        Dt = torch.zeros((code.N + 1, 4), dtype=torch.int32, device="cuda")
        itc = ctypes.c_int(0)
        check(lib.bldpc_decode(code._h, ctypes.c_void_p(yt.data_ptr()), 4, 5, 0, C.EXIT_PER_FRAME, 0, ctypes.c_void_p(Dt.data_ptr()), None, None,
                               ctypes.byref(itc), None), "bldpc_decode")


def test_per_frame_statistic_and_simulation_loop(C, orc):
    """Statistic with one iteration count per frame, and the Simulation_GPU loop in per-frame mode, against the oracle's
    Statistic fed frame by frame."""
    J, L, Z, F = 4, 24, 96, 64
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    seed = np.array([173, 173, 173], np.int32)
    y = orc.bldpc_awgn(seed, orc.bldpc_sigma(2.4), code.N, F)
    Dw, _, itw = _oracle_per_frame(orc, ocode, y, F, 50)
    cnt = np.zeros(5, np.int64)
    for f in range(F):  # Statistic on batches of one frame (Simulation.cu:245-262)
        orc.bldpc_statistic(cnt, f + 1, np.ascontiguousarray(Dw[:, f]), code.N, 1, code.K, int(itw[f]))
    from cuda_ldpc_amd.simulation import Simulation_GPU
    SIM = C.SimCounters()
    Simulation_GPU(code, np.array([173, 173, 173], np.int32), orc.bldpc_sigma(2.4), SIM, Num_Frames_OneTime=F, maxIT=50,
                   exit_mode=C.EXIT_PER_FRAME, max_batches=1, log=None)
    assert [SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration, SIM.num_False_Frames, SIM.num_Alarm_Frames] == list(cnt)
    assert SIM.Total_Iteration == int(itw.sum()) and SIM.num_Frames == F


def test_per_frame_exit_full_batch_properties(C, orc):
    """65 536 frames at the benchmark point: iteration counts within [1, max_iter], flag <=> stopped early or passing at the last
    iteration, and a sample of frames against the oracle."""
    J, L, Z, F = 4, 24, 96, 65536
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    y0 = _channel(orc, L * Z, 4096, 3.0).reshape(code.N, 4096)
    yt = torch.from_numpy(y0).cuda().repeat(1, F // 4096).contiguous()
    r = C.LDPC_Decoder_GPU(code, yt, max_iter=50, exit_mode=C.EXIT_PER_FRAME)
    torch.cuda.synchronize()
    it = r["iters"].cpu().numpy()
    flags = r["D"][code.N].cpu().numpy()
    assert it.min() >= 1 and it.max() <= 50
    assert np.all(flags[it < 50] == 1)
    assert np.array_equal(it[:4096], it[4096:8192]) and np.array_equal(it[:4096], it[-4096:])  # tiled input, tiled result
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    pick = [0, 1, 77, 4095]
    Dw, _, itw = _oracle_per_frame(orc, ocode, np.ascontiguousarray(y0[:, pick]), len(pick), 50)
    D = r["D"].cpu().numpy()
    for i, f in enumerate(pick):
        assert it[f + 8192] == itw[i] and np.array_equal(D[:, f + 8192], Dw[:, i])


@pytest.mark.parametrize("snr,length,s0,stop_want", [(-3.0, 4, 194, 7), (-3.0, 8, 203, 40), (-2.0, 4, 188, 40)])
def test_batch_global_when_a_frame_loses_its_flag_again(C, orc, snr, length, s0, stop_want):
    """The fused kernels find the reference's stop iteration from a per-frame pass (latest first flag m), then decode m
    iterations; when some frame is no longer flagged at m they must fall back to the full flag histories.  Short `length`
    (few examined bits) at low SNR makes flags come and go: cases found with the oracle."""
    J, L, Z, F = 4, 24, 96, 5
    y = _channel(orc, L * Z, F, snr, seed=(s0, 173, 173))
    ocode = orc.BinaryCode(_path(J, L, Z), J, L, Z)
    code = C.BinaryCode.from_blockh(_path(J, L, Z), J, L, Z)
    want = orc.bldpc_decode(ocode, y, F, 40, early_exit=1, length=length, want_app=True)
    assert want["it"] == stop_want
    for kern in KERNELS:
        got = _decode(C, code, y, F, max_iter=40, length=length, exit_mode=C.EXIT_BATCH_GLOBAL, kernel=_k(C, kern), want_app=True,
                      want_flag_hist=True)
        _assert_same(got, want, code.N, F)
        mask = np.uint64((1 << want["it"]) - 1)
        assert np.array_equal(got["flag_hist"] & mask, want["flag_hist"] & mask)


# ---- the persistent-workgroup instantiations (frames from per-XCD counters) are only reached by LARGE batches ------------------------
PERSIST_CODES = [("J4_L24_Z96_BlockH.txt", 4, 24, 96, 3.0, 8192, "halfrow"),     # k_qc2p: grid 4096 pairs > 512 resident workgroups
                 ("J32_L64_Z64_BlockH.txt", 32, 64, 64, -0.6, 4096, "row"),      # k_qcp
                 ("J10_L60_Z160_BlockH.txt", 10, 60, 160, 2.9, 2048, "compressed"),  # k_qcc<PERSIST>
                 ("PON_LDPC.txt", 12, 69, 256, 2.5, 2048, "regs"),               # k_qcr<PERSIST>
                 ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, -0.8, 1024, "regs")] # k_qcr2<PERSIST>


@pytest.mark.parametrize("fn,J,L,Z,snr,F,tag", PERSIST_CODES)
def test_persistent_kernels_equal_one_workgroup_per_frame_group(C, orc, monkeypatch, fn, J, L, Z, snr, F, tag):
    """Batches large enough for the persistent form of every fused tier (grid > resident workgroups): per-frame exit and the
    batch-global rule (whose pre-pass is a per-frame pass) give the same D, iteration counts and a-posteriori sums, bit for bit, as
    a code object created under BLDPC_NO_PERSIST=1 (one workgroup per frame group; the switch is read when the code is created),
    and a sample of frames equals the oracle's decode of that frame alone."""
    p = os.path.join(BL, fn)
    code = C.BinaryCode.from_blockh(p, J, L, Z)
    monkeypatch.setenv("BLDPC_NO_PERSIST", "1")
    plain = C.BinaryCode.from_blockh(p, J, L, Z)
    monkeypatch.delenv("BLDPC_NO_PERSIST")
    seed = np.array([173, 173, 173], np.int32)
    yt = C.AWGNChannel_GPU(seed, C.sigma_of(snr), code.N, F)
    a = C.LDPC_Decoder_GPU(code, yt, max_iter=50, exit_mode=C.EXIT_PER_FRAME, want_app=True)
    b = C.LDPC_Decoder_GPU(plain, yt, max_iter=50, exit_mode=C.EXIT_PER_FRAME, want_app=True)
    torch.cuda.synchronize()
    it = a["iters"].cpu().numpy()
    assert len(set(it.tolist())) > 3, "pick an SNR at which frames stop at different iterations"
    assert torch.equal(a["iters"], b["iters"]) and torch.equal(a["D"], b["D"])
    assert torch.equal(a["app"].view(torch.int32), b["app"].view(torch.int32))
    ocode = orc.BinaryCode(p, J, L, Z)
    pick = [0, 1, F // 2 + 1, F - 1]
    y = yt[:, pick].cpu().numpy()
    Dw, appw, itw = _oracle_per_frame(orc, ocode, np.ascontiguousarray(y), len(pick), 50)
    D, app = a["D"][:, pick].cpu().numpy(), a["app"][:, pick].cpu().numpy()
    assert np.array_equal(it[pick], itw) and np.array_equal(D, Dw) and np.array_equal(app.view(np.uint32), appw.view(np.uint32))
    # the reference's batch-global rule on the same batch: its pre-pass runs the persistent kernel too
    ga = C.LDPC_Decoder_GPU(code, yt, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL, want_app=True)
    gb = C.LDPC_Decoder_GPU(plain, yt, max_iter=50, exit_mode=C.EXIT_BATCH_GLOBAL, want_app=True)
    torch.cuda.synchronize()
    assert ga["iteraTime"] == gb["iteraTime"] and ga["iteraTime"] >= int(it.max())  # no earlier than the latest first flag
    assert torch.equal(ga["D"], gb["D"]) and torch.equal(ga["app"].view(torch.int32), gb["app"].view(torch.int32))
    w = orc.bldpc_decode(ocode, np.ascontiguousarray(y), len(pick), ga["iteraTime"], early_exit=0, want_app=True)  # the same iterations, fixed
    assert np.array_equal(ga["D"][:code.N, pick].cpu().numpy().reshape(-1), w["D"][:code.N * len(pick)])
    assert np.array_equal(ga["app"][:, pick].cpu().numpy().reshape(-1).view(np.uint32), w["app"].view(np.uint32))
    assert tag in code.last_kernel or "qc_lds" in code.last_kernel

