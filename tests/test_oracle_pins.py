"""Pin the CPU oracle before anything is compared against it (CPU only).

* binary path: PARITY UNPINNED (emulation only).  The D-hashes / anchor samples below are the ones SURVEY.md 8c + Appendix D.3
  recorded from a host emulation of the reference's kernels built with stand-in CUDA headers in the survey session; the reference
  has no buildable CPU path for the binary decoder, nothing here can regenerate them, and they cover J4_L24_Z96 and J32_L64_Z64
  hard bits only (no LLR bits, no other matrix).  tests/test_binary_crosscheck_cpu.py adds a second, independent restatement.
* NB EMS path: bit-exact against dumps of the REFERENCE's own CPU decoder
  (oracle/_ref/nb_ref built from /root/reference/myNBLDPC/src) committed under
  tests/golden/nb_ref_*.npz, and against the reference fixture codeword
  (include/codeword_test.h:1 -> data/nb/codeword_bds_gf64.txt).
"""
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN

BL = os.path.join(DATA, "bldpc")
NB = os.path.join(DATA, "nb")

# (file, J, L, Z, F, Es/N0, literal, expected hash of D[0..N*F), expected iteraTime)  -- SURVEY.md 8c
SURVEY_HASHES = [
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 32, 3.0, False, 0x05A41534, 50),
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 32, 4.0, False, 0x99F71DC5, 6),
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 32, 5.0, False, 0x43A7F222, 3),
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 32, 4.0, True, 0x90C5DF9B, 50),
    ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 16, -1.0, False, 0xC00D92C5, None),
    ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 16, 0.0, False, 0x5E509DC5, None),
]
# first sample y[0] of the first batch, J4_L24_Z96, F = 32
Y0_ANCHORS = {2.0: 0.848695815, 3.0: 0.865149975, 4.0: 0.879814804, 5.0: 0.892884851, 6.0: 0.904533505}
SIGMA_ANCHORS = {2.0: 0.561674893, 4.0: 0.446154207}


@pytest.mark.parametrize("case", SURVEY_HASHES, ids=lambda c: "%s_%g_%s" % (c[0][:11], c[5], "lit" if c[6] else "cor"))
def test_binary_oracle_matches_survey_hashes(orc, case):
    name, J, L, Z, F, snr, literal, want, want_it = case
    code = orc.BinaryCode(os.path.join(BL, name), J, L, Z, literal=literal)
    seed = np.array([173, 173, 173], np.int32)
    y = orc.bldpc_awgn(seed, orc.bldpc_sigma(snr), code.N, F)
    r = orc.bldpc_decode(code, y, F, 50, early_exit=1)
    assert orc.fold_hash(r["D"][: code.N * F]) == want
    if want_it is not None:
        assert r["it"] == want_it


def test_binary_channel_anchors(orc):
    for snr, want in Y0_ANCHORS.items():
        seed = np.array([173, 173, 173], np.int32)
        y = orc.bldpc_awgn(seed, orc.bldpc_sigma(snr), 2304, 32)
        assert np.float32(y[0]) == np.float32(want)
    for snr, want in SIGMA_ANCHORS.items():
        assert np.float32(orc.bldpc_sigma(snr)) == np.float32(want)


def test_binary_as_written_table_error_floor(orc):
    # SURVEY F3 / BASELINE.md 2: reference Transform_H as written -> 7/32 frame errors at 6 dB
    code = orc.BinaryCode(os.path.join(BL, "J4_L24_Z96_BlockH.txt"), 4, 24, 96, literal=True)
    seed = np.array([173, 173, 173], np.int32)
    y = orc.bldpc_awgn(seed, orc.bldpc_sigma(6.0), code.N, 32)
    r = orc.bldpc_decode(code, y, 32, 50, early_exit=1)
    assert int(r["D"][code.N * 32:].sum()) == 32 - 7


def test_binary_golden_fixture_consistent(orc):
    for fn in sorted(os.listdir(GOLDEN)):
        if not fn.startswith("bldpc_"):
            continue
        g = np.load(os.path.join(GOLDEN, fn))
        J, L, Z, F = int(g["J"]), int(g["L"]), int(g["Z"]), int(g["F"])
        code = orc.BinaryCode(os.path.join(BL, "J%d_L%d_Z%d_BlockH.txt" % (J, L, Z)), J, L, Z, literal=bool(g["literal"]))
        r = orc.bldpc_decode(code, g["y"], F, 50, early_exit=1)
        D = np.unpackbits(g["D_bits"])[: code.N * F].astype(np.int32)
        assert np.array_equal(r["D"][: code.N * F], D)
        assert np.array_equal(r["D"][code.N * F:], g["flags"])
        assert r["it"] == int(g["it"])
        assert orc.fold_hash(D) == int(g["hash"])
        # the stored y is what this box's libm regenerates (guards the channel restatement)
        seed = np.array([173, 173, 173], np.int32)
        y = orc.bldpc_awgn(seed, float(g["sigma"]), code.N, F)
        assert np.array_equal(y.view(np.uint32), g["y"].view(np.uint32))


# ----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def nbcode(orc):
    return orc.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))


def test_nb_fixture_codeword_is_valid(orc, nbcode):
    # codeword_test.h is a codeword of the BDS matrix (all 48 syndromes zero)
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    c = nbcode
    mul = c.mul.reshape(c.q, c.q)
    for row in range(c.M):
        s = 0
        for i in range(c.cn_w[row]):
            s ^= int(mul[cw[c.cn_vn[row * c.dc + i]], c.cn_gf[row * c.dc + i]])
        assert s == 0
    assert np.array_equal(c.add.reshape(c.q, c.q), np.arange(c.q)[:, None] ^ np.arange(c.q)[None, :])


@pytest.mark.parametrize("snr", [2, 3, 5])
def test_nb_oracle_bit_exact_vs_reference_dump(orc, nbcode, snr):
    g = np.load(os.path.join(GOLDEN, "nb_ref_%ddB.npz" % snr))
    c = nbcode
    sigma = float(g["sigma"])
    assert np.float32(orc.nb_sigma(float(g["snr"]), c.rate)) == np.float32(sigma)
    assert np.float32(c.rate) == np.float32(g["rate"])
    # channel restatement reproduces the reference's rx stream
    seed = np.array([173, 173, 173], np.int32)
    full = {int(f): i for i, f in enumerate(g["full_frames"])}
    for fr in range(g["rx"].shape[0]):
        rx, Lch = orc.nb_channel(c, g["cw"], seed, sigma)
        assert np.array_equal(rx.view(np.uint32), g["rx"][fr].view(np.uint32)), "rx frame %d" % fr
        Lch = orc.nb_demodulate(c, g["rx"][fr], sigma)
        assert orc.fold_hash(Lch) == int(g["Lch_hash"][fr])
        r = orc.nb_ems_decode(c, Lch, 2, 2, int(g["maxit"]), want_state=True)
        assert r["it"] == int(g["it"][fr]) and r["ok"] == int(g["ok"][fr]), "frame %d" % fr
        assert np.array_equal(r["out"], g["out"][fr])
        assert orc.fold_hash(r["LLR"]) == int(g["LLR_hash"][fr]), "LLR frame %d" % fr
        assert orc.fold_hash(r["c2v"]) == int(g["c2v_hash"][fr]), "c2v frame %d" % fr
        if fr in full:
            i = full[fr]
            assert np.array_equal(Lch.view(np.uint32), g["full_Lch"][i].view(np.uint32))
            assert np.array_equal(r["LLR"].view(np.uint32), g["full_LLR"][i].view(np.uint32))
            assert np.array_equal(r["c2v"].view(np.uint32), g["full_c2v"][i].view(np.uint32))


@pytest.mark.parametrize("snr", [2, 4, 6])
def test_nb_gf256_oracle_bit_exact_vs_reference_dump(orc, snr):
    """GF(256): the reference's own Decoding_EMS on its code LDPC_N96_K48_GF256_d1_exp.txt (oracle/_ref/nb_ref_gf256: define.h's
    Matrixfile / GFQ edited at build time; all-zero codeword) against the restatement -- channel stream, L_ch, symbols, iteration
    counts, return flags and the full final LLR / L_c2v bits of all 36 frames (1 ... 20 iterations deep)."""
    nbd = os.path.join(DATA, "nb")
    c = orc.NBCode(os.path.join(nbd, "LDPC_N96_K48_GF256_d1_exp.txt"), os.path.join(nbd, "GF", "Arith.Table.GF.256.txt"))
    assert (c.N, c.M, c.q, c.dv, c.dc) == (12, 6, 256, 2, 4)
    g = np.load(os.path.join(GOLDEN, "nb_ref_gf256_%ddB.npz" % snr))
    sigma = float(g["sigma"])
    assert np.float32(orc.nb_sigma(float(g["snr"]), c.rate)) == np.float32(sigma) and np.float32(c.rate) == np.float32(g["rate"])
    assert not g["cw"].any()
    seed = np.array([173, 173, 173], np.int32)
    for fr in range(g["rx"].shape[0]):
        rx, Lch = orc.nb_channel(c, g["cw"], seed, sigma)
        assert np.array_equal(rx.view(np.uint32), g["rx"][fr].view(np.uint32)), "rx frame %d" % fr
        assert np.array_equal(Lch.view(np.uint32), g["Lch"][fr].view(np.uint32)), "L_ch frame %d" % fr
        r = orc.nb_ems_decode(c, Lch, 2, 2, int(g["maxit"]), want_state=True)
        assert r["it"] == int(g["it"][fr]) and r["ok"] == int(g["ok"][fr]), "frame %d" % fr
        assert np.array_equal(r["out"], g["out"][fr])
        assert np.array_equal(r["LLR"].view(np.uint32), g["LLR"][fr].view(np.uint32)), "LLR frame %d" % fr
        assert np.array_equal(r["c2v"].view(np.uint32), g["c2v"][fr].view(np.uint32)), "c2v frame %d" % fr


@pytest.mark.parametrize("snr", [3, 5])
def test_nb_exponent_format_matrix_oracle_bit_exact_vs_reference_dump(orc, snr):
    """LDPC_N576_K288_GF64_d1_exp.txt stores exponents; the reference's Get_H reads them as field elements, two zero coefficients
    included (Simulation.cpp:347-467), and decodes that code.  The reference built with this Matrixfile (oracle/_ref/nb_ref_exp64,
    all-zero codeword) against the restatement: channel stream, symbols, iteration counts, flags, final LLR / L_c2v bits."""
    nbd = os.path.join(DATA, "nb")
    c = orc.NBCode(os.path.join(nbd, "LDPC_N576_K288_GF64_d1_exp.txt"), os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
    g = np.load(os.path.join(GOLDEN, "nb_ref_exp64_%ddB.npz" % snr))
    seed = np.array([173, 173, 173], np.int32)
    for fr in range(g["rx"].shape[0]):
        rx, Lch = orc.nb_channel(c, g["cw"], seed, float(g["sigma"]))
        assert np.array_equal(rx.view(np.uint32), g["rx"][fr].view(np.uint32)) and orc.fold_hash(Lch) == int(g["Lch_hash"][fr])
        r = orc.nb_ems_decode(c, Lch, 2, 2, int(g["maxit"]), want_state=True)
        assert r["it"] == int(g["it"][fr]) and r["ok"] == int(g["ok"][fr]) and np.array_equal(r["out"], g["out"][fr]), "frame %d" % fr
        assert np.array_equal(r["LLR"].view(np.uint32), g["LLR"][fr].view(np.uint32))
        assert np.array_equal(r["c2v"].view(np.uint32), g["c2v"][fr].view(np.uint32))


@pytest.mark.parametrize("snr", [11, 14])
def test_nb_qam64_branches_bit_exact_vs_reference_dump(orc, nbcode, snr):
    """The n_QAM != 2 branches (Modulate / AWGNChannel_CPU / Demodulate: LDPC_Encoder.cpp:18-68, LDPC_Decoder.cpp:160-169) against the
    reference itself built with n_QAM 64 and Constellation/GRAY_64QAM.txt (oracle/_ref/nb_ref_qam64): the complex channel samples,
    L_ch bits and the decode results of 16 frames.  (Round 1 had these branches "parity unpinned".)"""
    c = nbcode
    con = orc.nb_read_constellation(os.path.join(NB, "Constellation", "GRAY_64QAM.txt"), 64)
    g = np.load(os.path.join(GOLDEN, "nb_ref_qam64_%ddB.npz" % snr))
    sigma = float(g["sigma"])
    assert np.float32(orc.nb_sigma(float(g["snr"]), c.rate, 0, 64)) == np.float32(sigma)
    seed = np.array([173, 173, 173], np.int32)
    for fr in range(g["rx"].shape[0]):
        rx, Lch = orc.nb_channel_qam(c, g["cw"], seed, sigma, con)
        assert np.array_equal(np.asarray(rx).reshape(-1, 2).view(np.uint32), g["rx"][fr].view(np.uint32)), "rx frame %d" % fr
        assert np.array_equal(Lch.view(np.uint32), g["Lch"][fr].view(np.uint32)), "L_ch frame %d" % fr
        r = orc.nb_ems_decode(c, Lch, 2, 2, int(g["maxit"]), want_state=True)
        assert r["it"] == int(g["it"][fr]) and r["ok"] == int(g["ok"][fr]) and np.array_equal(r["out"], g["out"][fr])
        assert orc.fold_hash(r["LLR"]) == int(g["LLR_hash"][fr]) and orc.fold_hash(r["c2v"]) == int(g["c2v_hash"][fr])


@pytest.mark.parametrize("snr", [14, 18])
def test_nb_gf256_qam256_bit_exact_vs_reference_dump(orc, snr):
    """GF(256) code over Gray 256-QAM, one point per symbol: the reference built with GFQ 256 / n_QAM 256
    (oracle/_ref/nb_ref_gf256_qam256, all-zero codeword) against the restatement: channel samples, L_ch, decode results."""
    nbd = os.path.join(DATA, "nb")
    c = orc.NBCode(os.path.join(nbd, "LDPC_N96_K48_GF256_d1_exp.txt"), os.path.join(nbd, "GF", "Arith.Table.GF.256.txt"))
    con = orc.nb_read_constellation(os.path.join(nbd, "Constellation", "GRAY_256QAM.txt"), 256)
    g = np.load(os.path.join(GOLDEN, "nb_ref_gf256_qam256_%ddB.npz" % snr))
    sigma = float(g["sigma"])
    assert np.float32(orc.nb_sigma(float(g["snr"]), c.rate, 0, 256)) == np.float32(sigma)
    seed = np.array([173, 173, 173], np.int32)
    for fr in range(g["rx"].shape[0]):
        rx, Lch = orc.nb_channel_qam(c, g["cw"], seed, sigma, con)
        assert np.array_equal(np.asarray(rx).reshape(-1, 2).view(np.uint32), g["rx"][fr].view(np.uint32)), "rx frame %d" % fr
        assert np.array_equal(Lch.view(np.uint32), g["Lch"][fr].view(np.uint32)), "L_ch frame %d" % fr
        r = orc.nb_ems_decode(c, Lch, 2, 2, int(g["maxit"]), want_state=True)
        assert r["it"] == int(g["it"][fr]) and r["ok"] == int(g["ok"][fr]) and np.array_equal(r["out"], g["out"][fr])
        assert orc.fold_hash(r["LLR"]) == int(g["LLR_hash"][fr]) and orc.fold_hash(r["c2v"]) == int(g["c2v_hash"][fr])


HEAVY = {"tanner16": ("Tanner_74_9_Z128_GF16.txt", "Arith.Table.GF.16.txt", (9472, 1152, 16, 3, 21)),
         "gf256_dc12": ("LDPC_N576_K480_GF256_exp.txt", "Arith.Table.GF.256.txt", (72, 12, 256, 2, 12))}


@pytest.mark.parametrize("tag,snr", [("tanner16", 5), ("tanner16", 6), ("gf256_dc12", 5), ("gf256_dc12", 7)])
def test_nb_heavy_row_codes_oracle_bit_exact_vs_reference_dump(orc, tag, snr):
    """The reference's two codes with check rows heavier than 6 -- Tanner_74_9_Z128_GF16.txt (row weight 21, 9472 symbols) and
    LDPC_N576_K480_GF256_exp.txt (GF(256), row weight 12) -- through its own Decoding_EMS (oracle/_ref/nb_ref_tanner16,
    nb_ref_gf256_dc12: define.h's Matrixfile / GFQ / maxdc / maxdv edited at build time, all-zero codeword) against the restatement:
    channel stream, symbols, iteration counts (0 ... 20), flags, and hashes of the L_ch / final LLR / L_c2v bits."""
    mat, tab, dims = HEAVY[tag]
    nbd = os.path.join(DATA, "nb")
    c = orc.NBCode(os.path.join(nbd, mat), os.path.join(nbd, "GF", tab))
    assert (c.N, c.M, c.q, c.dv, c.dc) == dims
    g = np.load(os.path.join(GOLDEN, "nb_ref_%s_%ddB.npz" % (tag, snr)))
    sigma = float(g["sigma"])
    assert np.float32(orc.nb_sigma(float(g["snr"]), c.rate)) == np.float32(sigma) and np.float32(c.rate) == np.float32(g["rate"])
    cw = g["cw"].astype(np.int32)
    seed = np.array([173, 173, 173], np.int32)
    for fr in range(g["rx"].shape[0]):
        rx, Lch = orc.nb_channel(c, cw, seed, sigma)
        assert np.array_equal(rx.view(np.uint32), g["rx"][fr].view(np.uint32)), "rx frame %d" % fr
        assert orc.fold_hash(Lch) == int(g["Lch_hash"][fr])
        r = orc.nb_ems_decode(c, Lch, 2, 2, int(g["maxit"]), want_state=True)
        assert r["it"] == int(g["it"][fr]) and r["ok"] == int(g["ok"][fr]), "frame %d" % fr
        assert np.array_equal(r["out"], g["out"][fr].astype(np.int32))
        assert orc.fold_hash(r["LLR"]) == int(g["LLR_hash"][fr]) and orc.fold_hash(r["c2v"]) == int(g["c2v_hash"][fr]), "frame %d" % fr


@pytest.mark.parametrize("snr", [2, 3, 5])
@pytest.mark.parametrize("layered", [False, True])
def test_nb_tmm_oracle_bit_exact_vs_reference_dump(orc, nbcode, snr, layered):
    """Decoding_TMM / Decoding_layered_TMM (decoder_method 1 / 3): the restatement against the reference's own
    functions (oracle/_ref/nb_ref dump ... 1|3), 16 frames per Eb/N0: symbols, iteration counts, return flags and the
    final LLR / L_c2v state bit for bit."""
    g = np.load(os.path.join(GOLDEN, "nb_ref_%s_%ddB.npz" % ("ltmm" if layered else "tmm", snr)))
    c = nbcode
    full = {int(f): i for i, f in enumerate(g["full_frames"])}
    for fr in range(g["rx"].shape[0]):
        Lch = orc.nb_demodulate(c, g["rx"][fr], float(g["sigma"]))
        assert orc.fold_hash(Lch) == int(g["Lch_hash"][fr])
        r = orc.nb_tmm_decode(c, Lch, int(g["maxit"]), layered=layered, want_state=True)
        assert r["it"] == int(g["it"][fr]) and r["ok"] == int(g["ok"][fr]), "frame %d" % fr
        assert np.array_equal(r["out"], g["out"][fr])
        assert orc.fold_hash(r["LLR"]) == int(g["LLR_hash"][fr]), "LLR frame %d" % fr
        assert orc.fold_hash(r["c2v"]) == int(g["c2v_hash"][fr]), "c2v frame %d" % fr
        if fr in full:
            i = full[fr]
            assert np.array_equal(r["LLR"].view(np.uint32), g["full_LLR"][i].view(np.uint32))
            assert np.array_equal(r["c2v"].view(np.uint32), g["full_c2v"][i].view(np.uint32))


def test_nb_survey_anchor_values(orc, nbcode):
    # 3 dB, frame 0..3, L_ch of symbol 0 element 1; iteration counts
    g = np.load(os.path.join(GOLDEN, "nb_ref_3dB.npz"))
    assert np.float32(g["sigma"]) == np.float32(0.707945764)
    assert list(g["it"][:8]) == [3, 4, 5, 7, 20, 20, 8, 12]
    assert list(g["ok"][:8]) == [1, 1, 1, 1, 0, 0, 1, 1]
    assert np.float32(g["full_Lch"][0][0, 0]) == np.float32(-3.2869091)
    assert int((g["out"][4] != g["cw"]).sum()) == 9 and int((g["out"][5] != g["cw"]).sum()) == 63


@pytest.mark.skipif(not os.path.isdir("/root/reference/myNBLDPC"), reason="reference tree absent (GPU box)")
def test_nb_reference_binary_reproduces_golden(orc):
    """Re-run the real reference here and compare with the committed dump (guards fixture drift)."""
    import subprocess
    import tempfile
    import importlib.util
    ref = orc.ref_binary()
    if ref is None:
        pytest.skip("oracle/_ref/nb_ref not built")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLDEN, "nb_ref_3dB.npz"))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.bin")
        subprocess.check_call([ref, "dump", "3.0", "8", out], cwd=NB, stdout=subprocess.DEVNULL)
        d = mg.parse_nb_dump(out)
    for fr, r in enumerate(d["recs"]):
        assert np.array_equal(r["out"], g["out"][fr]) and r["it"] == int(g["it"][fr])
        assert orc.fold_hash(r["LLR"]) == int(g["LLR_hash"][fr])
