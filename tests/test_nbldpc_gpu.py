"""GPU parity tests of the GF(q) EMS decoder (through the C ABI, include/nbldpc.h).

Bit-exact (symbols, iteration counts, return flags, and the float LLR / c2v state compared as uint32)
against (1) dumps of the REFERENCE's own CPU decoder committed under tests/golden/nb_ref_*.npz and
(2) the CPU oracle on further seeded inputs.
"""
import os

import numpy as np
import pytest
import torch

from conftest import DATA, GOLDEN

pytestmark = pytest.mark.gpu
NB = os.path.join(DATA, "nb")


@pytest.fixture(scope="module")
def nb():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cuda_ldpc_amd import nbldpc
    return nbldpc


@pytest.fixture(scope="module")
def code(nb):
    mul, _, _ = nb.GFInitial(64, os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    return nb.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), mul)


@pytest.fixture(scope="module")
def ocode(orc):
    return orc.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))


@pytest.mark.parametrize("snr", [2, 3, 5])
def test_matches_reference_dump(nb, code, orc, snr):
    """Inputs = the reference's channel samples; outputs vs the reference's own Decoding_EMS."""
    g = np.load(os.path.join(GOLDEN, "nb_ref_%ddB.npz" % snr))
    sigma = float(g["sigma"])
    rx = torch.from_numpy(g["rx"]).cuda()
    Lch = nb.Demodulate(code, rx, sigma)
    r = nb.Decoding_EMS(code, Lch, 2, 2, int(g["maxit"]), want_state=True)
    torch.cuda.synchronize()
    Lch_h = Lch.cpu().numpy()
    out, it, ok = r["DecodeOutput"].cpu().numpy(), r["iter_number"].cpu().numpy(), r["ok"].cpu().numpy()
    LLR, c2v = r["LLR"].cpu().numpy(), r["L_c2v"].cpu().numpy()
    assert np.array_equal(it, g["it"]) and np.array_equal(ok, g["ok"])
    assert np.array_equal(out, g["out"])
    for fr in range(rx.shape[0]):
        assert orc.fold_hash(Lch_h[fr]) == int(g["Lch_hash"][fr]), "L_ch frame %d" % fr
        assert orc.fold_hash(LLR[fr]) == int(g["LLR_hash"][fr]), "LLR frame %d" % fr
        assert orc.fold_hash(c2v[fr]) == int(g["c2v_hash"][fr]), "c2v frame %d" % fr
    for i, fr in enumerate(g["full_frames"]):
        assert np.array_equal(LLR[fr].view(np.uint32), g["full_LLR"][i].view(np.uint32))
        assert np.array_equal(c2v[fr].view(np.uint32), g["full_c2v"][i].view(np.uint32))


@pytest.mark.parametrize("iters", [1, 2, 3, 6])
def test_per_iteration_state_vs_oracle(nb, code, ocode, orc, iters):
    """Per-iteration LLR snapshots: run with maxIT = k and compare the whole float state bitwise."""
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    seed = np.array([173, 173, 173], np.int32)
    sigma = nb.sigma_of(2.5, code.rate)
    rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw) for _ in range(6)])
    Lch = nb.Demodulate(code, torch.from_numpy(rx).cuda(), sigma)
    r = nb.Decoding_EMS(code, Lch, 2, 2, iters, want_state=True)
    torch.cuda.synchronize()
    for b in range(rx.shape[0]):
        want = orc.nb_ems_decode(ocode, orc.nb_demodulate(ocode, rx[b], sigma), 2, 2, iters, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))


@pytest.mark.parametrize("Nm,Nc", [(2, 2), (3, 2), (2, 3), (4, 1), (1, 1)])
def test_other_ems_parameters(nb, code, ocode, orc, Nm, Nc):
    """EMS_NM / EMS_NC other than the default, including EMS_NC == maxdc-1 (-> Nc = w-1, LDPC_Decoder.cpp:294)."""
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    seed = np.array([11, 22, 33], np.int32)
    sigma = nb.sigma_of(3.0, code.rate)
    rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw) for _ in range(3)])
    Lch = nb.Demodulate(code, torch.from_numpy(rx).cuda(), sigma)
    r = nb.Decoding_EMS(code, Lch, Nm, Nc, 8, want_state=True)
    torch.cuda.synchronize()
    for b in range(rx.shape[0]):
        want = orc.nb_ems_decode(ocode, orc.nb_demodulate(ocode, rx[b], sigma), Nm, Nc, 8, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))


def test_ties_zeros_and_extremes(nb, code, ocode, orc):
    """Inputs full of exact ties, zeros and +-0: the stable sort order and first-max decision must still match."""
    rng = np.random.default_rng(3)
    B = 4
    Lch = rng.integers(-3, 4, size=(B, code.N, code.q - 1)).astype(np.float32)   # heavy ties
    Lch[0, :, ::5] = -0.0
    Lch[1] *= 1e30
    Lch[2] *= 1e-40  # denormals
    r = nb.Decoding_EMS(code, torch.from_numpy(Lch).cuda(), 2, 2, 5, want_state=True)
    torch.cuda.synchronize()
    for b in range(B):
        want = orc.nb_ems_decode(ocode, Lch[b], 2, 2, 5, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))


def test_values_that_differ_only_in_their_last_bits(nb, code, ocode, orc):
    """The sort of phase B runs on 32-bit keys (value image without its low 6 bits | 63 - index), verifies the permutation against
    the full (value, index) order and repeats a vector on the 64-bit keys when two values differ in nothing but those 6 bits.  Here
    every channel vector is a handful of neighbouring floats (a few ulps apart, both signs), so nearly every vector takes the
    repeat path: symbols, iteration counts, LLR and c2v bits must still equal the oracle's."""
    rng = np.random.default_rng(11)
    B = 6
    base = np.array([1.0, -1.0, 0.37, -2.5, 1e-3, 7.0], np.float32)
    Lch = np.empty((B, code.N, code.q - 1), np.float32)
    for b in range(B):
        ulps = rng.integers(0, 40, size=(code.N, code.q - 1)).astype(np.int32)
        v = np.full((code.N, code.q - 1), base[b], np.float32).view(np.int32) + ulps  # same value, 0..39 ulps away
        Lch[b] = v.view(np.float32)
    Lch[3, ::2] *= -1.0
    r = nb.Decoding_EMS(code, torch.from_numpy(Lch).cuda(), 2, 2, 4, want_state=True)
    torch.cuda.synchronize()
    for b in range(B):
        want = orc.nb_ems_decode(ocode, Lch[b], 2, 2, 4, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))


@pytest.mark.parametrize("snr", [2, 4, 6])
def test_gf256_matches_reference_dump(nb, orc, snr):
    """GF(256), the reference's code LDPC_N96_K48_GF256_d1_exp.txt (12 symbols, 6 checks): a message vector spans four waves
    (k_nb_ems_wide).  Inputs = the reference's channel samples; Demodulate, symbols, iteration counts, return flags and the final
    LLR / L_c2v bits vs the reference's own Decoding_EMS built for GFQ 256 (tests/golden/nb_ref_gf256_*.npz, 36 frames)."""
    mul, _, _ = nb.GFInitial(256, os.path.join(NB, "GF", "Arith.Table.GF.256.txt"))
    code = nb.NBCode(os.path.join(NB, "LDPC_N96_K48_GF256_d1_exp.txt"), mul)
    assert (code.N, code.M, code.q) == (12, 6, 256)
    g = np.load(os.path.join(GOLDEN, "nb_ref_gf256_%ddB.npz" % snr))
    sigma = float(g["sigma"])
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), sigma)
    assert np.array_equal(Lch.cpu().numpy().view(np.uint32), g["Lch"].view(np.uint32))
    r = nb.Decoding_EMS(code, Lch, 2, 2, int(g["maxit"]), want_state=True)
    torch.cuda.synchronize()
    assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["ok"].cpu().numpy(), g["ok"])
    assert np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"])
    assert np.array_equal(r["LLR"].cpu().numpy().view(np.uint32), g["LLR"].view(np.uint32))
    assert np.array_equal(r["L_c2v"].cpu().numpy().view(np.uint32), g["c2v"].view(np.uint32))


def test_gf256_ties_and_batch_independence(nb, orc):
    """GF(256) on inputs full of exact ties and signed zeros (first-maximum rule and stable order across the four waves of a
    vector) against the oracle, and a 2 048-frame batch whose tiles must decode identically."""
    nbd = NB
    mul, _, _ = nb.GFInitial(256, os.path.join(nbd, "GF", "Arith.Table.GF.256.txt"))
    code = nb.NBCode(os.path.join(nbd, "LDPC_N96_K48_GF256_d1_exp.txt"), mul)
    ocode = orc.NBCode(os.path.join(nbd, "LDPC_N96_K48_GF256_d1_exp.txt"), os.path.join(nbd, "GF", "Arith.Table.GF.256.txt"))
    rng = np.random.default_rng(5)
    B = 4
    Lch = rng.integers(-2, 3, size=(B, code.N, code.q - 1)).astype(np.float32)
    Lch[0, :, ::7] = -0.0
    Lch[1] *= 1e30
    r = nb.Decoding_EMS(code, torch.from_numpy(Lch).cuda(), 2, 2, 3, want_state=True)
    torch.cuda.synchronize()
    for b in range(B):
        want = orc.nb_ems_decode(ocode, Lch[b], 2, 2, 3, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))
    g = np.load(os.path.join(GOLDEN, "nb_ref_gf256_4dB.npz"))
    Lt = torch.from_numpy(g["Lch"][:8]).cuda().repeat(256, 1, 1).contiguous()
    r = nb.Decoding_EMS(code, Lt, 2, 2, int(g["maxit"]))
    torch.cuda.synchronize()
    out = r["DecodeOutput"].view(256, 8, code.N)
    assert bool((out == out[:1]).all()) and np.array_equal(out[0].cpu().numpy(), g["out"][:8])
    assert np.array_equal(r["iter_number"].view(256, 8)[0].cpu().numpy(), g["it"][:8])


@pytest.mark.parametrize("snr", [3, 5])
def test_exponent_format_matrix_matches_reference_dump(nb, snr):
    """LDPC_N576_K288_GF64_d1_exp.txt read the way the reference's Get_H reads it (exponents as field elements, two zero
    coefficients included): EMS against the reference's own decoder built with this Matrixfile; the trellis decoders, which need
    the inverse of every coefficient (GFInverse(0) exits in the reference), refuse the code."""
    mul, _, _ = nb.GFInitial(64, os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    code = nb.NBCode(os.path.join(NB, "LDPC_N576_K288_GF64_d1_exp.txt"), mul)
    g = np.load(os.path.join(GOLDEN, "nb_ref_exp64_%ddB.npz" % snr))
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), float(g["sigma"]))
    r = nb.Decoding_EMS(code, Lch, 2, 2, int(g["maxit"]), want_state=True)
    torch.cuda.synchronize()
    assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["ok"].cpu().numpy(), g["ok"])
    assert np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"])
    assert np.array_equal(r["LLR"].cpu().numpy().view(np.uint32), g["LLR"].view(np.uint32))
    assert np.array_equal(r["L_c2v"].cpu().numpy().view(np.uint32), g["c2v"].view(np.uint32))
    with pytest.raises(Exception):
        nb.Decoding_TMM(code, Lch, int(g["maxit"]))


@pytest.mark.parametrize("snr", [14, 18])
def test_gf256_qam256_matches_reference_dump(nb, orc, snr):
    """GF(256) code over Gray 256-QAM (fields and constellations beyond the north star) against the reference built with
    GFQ 256 / n_QAM 256: host channel samples, device Demodulate L_ch bits, k_nb_ems_wide's symbols / iteration counts / state hashes."""
    mul, _, _ = nb.GFInitial(256, os.path.join(NB, "GF", "Arith.Table.GF.256.txt"))
    code = nb.NBCode(os.path.join(NB, "LDPC_N96_K48_GF256_d1_exp.txt"), mul)
    con = nb.Get_CONSTELLATION(os.path.join(NB, "Constellation", "GRAY_256QAM.txt"), 256)
    g = np.load(os.path.join(GOLDEN, "nb_ref_gf256_qam256_%ddB.npz" % snr))
    sigma = float(g["sigma"])
    seed = np.array([173, 173, 173], np.int32)
    rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, g["cw"], CONSTELLATION=con) for _ in range(g["rx"].shape[0])])
    assert np.array_equal(rx.view(np.uint32), g["rx"].view(np.uint32))
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), sigma, CONSTELLATION=torch.from_numpy(con).cuda())
    assert np.array_equal(Lch.cpu().numpy().view(np.uint32), g["Lch"].view(np.uint32))
    r = nb.Decoding_EMS(code, Lch, 2, 2, int(g["maxit"]), want_state=True)
    torch.cuda.synchronize()
    assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["ok"].cpu().numpy(), g["ok"])
    assert np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"])
    for fr in range(g["rx"].shape[0]):
        assert orc.fold_hash(r["LLR"][fr].cpu().numpy()) == int(g["LLR_hash"][fr])
        assert orc.fold_hash(r["L_c2v"][fr].cpu().numpy()) == int(g["c2v_hash"][fr])


HEAVY = {"tanner16": ("Tanner_74_9_Z128_GF16.txt", "Arith.Table.GF.16.txt", 16), "gf256_dc12": ("LDPC_N576_K480_GF256_exp.txt", "Arith.Table.GF.256.txt", 256)}


@pytest.mark.parametrize("tag,snr", [("tanner16", 5), ("tanner16", 6), ("gf256_dc12", 5), ("gf256_dc12", 7)])
def test_heavy_row_codes_match_reference_dump(nb, orc, tag, snr):
    """The reference's codes with check rows heavier than 6 (Tanner_74_9_Z128_GF16.txt: 9472 symbols, row weight 21;
    LDPC_N576_K480_GF256_exp.txt: GF(256), row weight 12) go through k_nb_ems_hbm (state in a global-memory workspace, the
    reference's recursion executed as is): symbols, iteration counts (0 ... 20), flags and the L_ch / final LLR / L_c2v bits (hashes)
    against the reference's own Decoding_EMS built for these files (tests/golden/nb_ref_tanner16_*, nb_ref_gf256_dc12_*)."""
    mat, tab, q = HEAVY[tag]
    mul, _, _ = nb.GFInitial(q, os.path.join(NB, "GF", tab))
    code = nb.NBCode(os.path.join(NB, mat), mul)
    g = np.load(os.path.join(GOLDEN, "nb_ref_%s_%ddB.npz" % (tag, snr)))
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), float(g["sigma"]))
    r = nb.Decoding_EMS(code, Lch, 2, 2, int(g["maxit"]), want_state=True)
    torch.cuda.synchronize()
    assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["ok"].cpu().numpy(), g["ok"])
    assert np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"].astype(np.int32))
    for fr in range(g["rx"].shape[0]):
        assert orc.fold_hash(Lch[fr].cpu().numpy()) == int(g["Lch_hash"][fr])
        assert orc.fold_hash(r["LLR"][fr].cpu().numpy()) == int(g["LLR_hash"][fr]), "LLR frame %d" % fr
        assert orc.fold_hash(r["L_c2v"][fr].cpu().numpy()) == int(g["c2v_hash"][fr]), "c2v frame %d" % fr


def test_workspace_kernel_equals_fused_kernels_and_reference(nb, code, orc, monkeypatch):
    """k_nb_ems_hbm forced (NBLDPC_FORCE_HBM) onto codes the fused kernels take: the BDS GF(64) code against the reference's dumps at
    three Eb/N0 and against k_nb_ems for other (Nm, Nc) -- the `Nc == maxdc - 1` rule of LDPC_Decoder.cpp:294 included; the
    exponent-format file with its zero coefficients; ties and signed zeros on the GF(256) row-weight-12 code against the oracle;
    and a batch larger than the number of workspace slots (a workgroup then decodes several frames, one after the other)."""
    monkeypatch.setenv("NBLDPC_FORCE_HBM", "1")
    mul, _, _ = nb.GFInitial(64, os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    forced = nb.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), mul)
    forced_exp = nb.NBCode(os.path.join(NB, "LDPC_N576_K288_GF64_d1_exp.txt"), mul)
    monkeypatch.delenv("NBLDPC_FORCE_HBM")
    for snr in (2, 3, 5):
        g = np.load(os.path.join(GOLDEN, "nb_ref_%ddB.npz" % snr))
        Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), float(g["sigma"]))
        r = nb.Decoding_EMS(forced, Lch, 2, 2, int(g["maxit"]), want_state=True)
        torch.cuda.synchronize()
        assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["ok"].cpu().numpy(), g["ok"])
        assert np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"])
        for fr in range(g["rx"].shape[0]):
            assert orc.fold_hash(r["LLR"][fr].cpu().numpy()) == int(g["LLR_hash"][fr])
            assert orc.fold_hash(r["L_c2v"][fr].cpu().numpy()) == int(g["c2v_hash"][fr])
        for Nm, Nc, maxdc in ((4, 1, 0), (3, 3, 4), (2, 5, 6), (64, 1, 0)):
            a = nb.Decoding_EMS(code, Lch, Nm, Nc, 6, maxdc=maxdc, want_state=True)
            b = nb.Decoding_EMS(forced, Lch, Nm, Nc, 6, maxdc=maxdc, want_state=True)
            torch.cuda.synchronize()
            for key in ("DecodeOutput", "iter_number", "ok"):
                assert torch.equal(a[key], b[key]), (Nm, Nc, key)
            for key in ("LLR", "L_c2v"):
                assert torch.equal(a[key].view(torch.int32), b[key].view(torch.int32)), (Nm, Nc, key)
    g = np.load(os.path.join(GOLDEN, "nb_ref_exp64_3dB.npz"))
    Lch = nb.Demodulate(forced_exp, torch.from_numpy(g["rx"]).cuda(), float(g["sigma"]))
    r = nb.Decoding_EMS(forced_exp, Lch, 2, 2, int(g["maxit"]), want_state=True)
    torch.cuda.synchronize()
    assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"])
    assert np.array_equal(r["LLR"].cpu().numpy().view(np.uint32), g["LLR"].view(np.uint32))
    assert np.array_equal(r["L_c2v"].cpu().numpy().view(np.uint32), g["c2v"].view(np.uint32))
    # ties and signed zeros, GF(256), row weight 12
    mul256, _, _ = nb.GFInitial(256, os.path.join(NB, "GF", "Arith.Table.GF.256.txt"))
    c12 = nb.NBCode(os.path.join(NB, "LDPC_N576_K480_GF256_exp.txt"), mul256)
    o12 = orc.NBCode(os.path.join(NB, "LDPC_N576_K480_GF256_exp.txt"), os.path.join(NB, "GF", "Arith.Table.GF.256.txt"))
    rng = np.random.default_rng(11)
    L = rng.integers(-2, 3, size=(3, c12.N, c12.q - 1)).astype(np.float32)
    L[0, :, ::5] = -0.0
    L[1] *= 1e30
    r = nb.Decoding_EMS(c12, torch.from_numpy(L).cuda(), 2, 2, 2, want_state=True)
    torch.cuda.synchronize()
    for b in range(3):
        want = orc.nb_ems_decode(o12, L[b], 2, 2, 2, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))
    # more frames than workspace slots (1024)
    g = np.load(os.path.join(GOLDEN, "nb_ref_gf256_dc12_5dB.npz"))
    Lt = nb.Demodulate(c12, torch.from_numpy(g["rx"]).cuda().repeat(300, 1).contiguous(), float(g["sigma"]))
    r = nb.Decoding_EMS(c12, Lt, 2, 2, int(g["maxit"]))
    torch.cuda.synchronize()
    out = r["DecodeOutput"].view(300, 4, c12.N)
    assert bool((out == out[:1]).all()) and np.array_equal(out[0].cpu().numpy(), g["out"].astype(np.int32))
    assert np.array_equal(r["iter_number"].view(300, 4)[7].cpu().numpy(), g["it"])


def test_full_size_batch_properties(nb, code, ocode, orc):
    """BASELINE config 5 size (16384 frames): a 32-frame oracle-checked block tiled 512 times; every tile must
    decode identically (frames are independent) and transmitted codewords that decode must be codewords."""
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    seed = np.array([173, 173, 173], np.int32)
    sigma = nb.sigma_of(3.0, code.rate)
    rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw) for _ in range(32)])
    rxt = torch.from_numpy(rx).cuda().repeat(512, 1).contiguous()
    Lch = nb.Demodulate(code, rxt, sigma)
    r = nb.Decoding_EMS(code, Lch, 2, 2, 20)
    torch.cuda.synchronize()
    out = r["DecodeOutput"].view(512, 32, code.N)
    assert bool((out == out[:1]).all())
    it = r["iter_number"].view(512, 32)
    assert bool((it == it[:1]).all())
    want = orc.nb_ems_decode_batch(ocode, np.stack([orc.nb_demodulate(ocode, rx[b], sigma) for b in range(32)]), 2, 2, 20)
    assert np.array_equal(out[0].cpu().numpy(), want["out"])
    assert np.array_equal(it[0].cpu().numpy(), want["it"])
    okm = r["ok"].view(512, 32)[0].cpu().numpy().astype(bool)
    assert np.array_equal(okm.astype(np.int32), want["ok"])
    # zero syndrome on success
    mul = ocode.mul.reshape(64, 64)
    o = out[0].cpu().numpy()
    for b in np.nonzero(okm)[0]:
        for row in range(code.M):
            s = 0
            for i in range(code.cn_weight[row]):
                s ^= int(mul[o[b, code.cn_linkVNs[row, i]], code.cn_linkVNs_GF[row, i]])
            assert s == 0
    # statistics kernel
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    nb.Statistic(code, counters, r, torch.from_numpy(cw).cuda())
    c = counters.cpu().tolist()
    errs = (o != cw[None, :]).sum(axis=1)
    assert c[0] == 512 * int((errs != 0).sum()) and c[1] == 512 * int(errs.sum()) and c[2] == 512 * int(want["it"].sum())


def test_argument_errors(nb, code):
    with pytest.raises(Exception):
        nb.Decoding_EMS(code, torch.zeros((2, code.N, 10), device="cuda"))
    with pytest.raises(Exception):
        nb.Decoding_EMS(code, torch.zeros((2, code.N, 63), device="cuda"), maxIT=0)
    with pytest.raises(Exception):
        nb.GFInitial(64, "/nonexistent")


def test_nb_simulation_counts_match_reference_style_loop(nb, code, ocode, orc):
    """Simulation_GPU (NB): batch decoding accounted in stream order stops exactly where the reference's per-frame loop does."""
    from cuda_ldpc_amd.nb_simulation import NBSim, Simulation_GPU
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    snr = 2.0
    sigma = nb.sigma_of(snr, code.rate)
    seed = np.array([173, 173, 173], np.int32)
    SIM = NBSim(snr)
    stop = Simulation_GPU(code, seed, sigma, SIM, cw, batch=64, leastErrorFrames=8, leastTestFrames=20)
    assert stop == 1
    # CPU replay of decode_once_cpu (Simulation.cpp:41-83) with the oracle
    oseed = np.array([173, 173, 173], np.int32)
    frames = errf = errb = its = 0
    while errf < 8 or frames < 20:
        rx, Lch = orc.nb_channel(ocode, cw, oseed, sigma)
        r = orc.nb_ems_decode(ocode, Lch, 2, 2, 20)
        frames += 1
        its += r["it"]
        e = int((r["out"] != cw).sum())
        errb += e
        errf += 1 if e else 0
    assert (SIM.num_Frames, SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration) == (frames, errf, errb, its)
    assert np.array_equal(seed, oseed)



def test_bitonic_steps(tmp_path):
    """Every compare-exchange step (K, J) of the kernel's 64-lane bitonic network -- DPP-fused for partners 1, 2 and 8
    lanes away, LDS crossbar for the rest -- against a plain 64-bit compare, on duplicated and distinct keys."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dpp_step_test")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function",
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "dpp_step_test.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "total 0" in out.stdout, out.stdout + out.stderr


def test_device_channel_same_draws_as_host(nb, code):
    """nbldpc_awgn_channel_device: LCG jump-ahead reproduces the serial stream of B frames; only the device libm may
    move a sample by an ulp.  Seeds after the batch are identical to B host calls."""
    cw = np.loadtxt(os.path.join(DATA, "nb", "codeword_bds_gf64.txt"), dtype=np.int32)
    sigma = nb.sigma_of(3.0, code.rate)
    s_host = np.array([173, 173, 173], np.int32)
    s_dev = s_host.copy()
    B = 9
    rx_h = np.stack([nb.AWGNChannel_CPU(s_host, sigma, code, cw) for _ in range(B)])
    rx_d = nb.AWGNChannel_GPU(s_dev, sigma, code, torch.from_numpy(cw).cuda(), B).cpu().numpy()
    assert np.array_equal(s_host, s_dev)
    same = (rx_h.view(np.int32) == rx_d.view(np.int32)).mean()
    assert same > 0.85 and np.abs(rx_h - rx_d).max() < 1e-6, (same, np.abs(rx_h - rx_d).max())  # measured: 89.4 % bit-identical, rest 1 ulp
    assert np.array_equal(nb.seed_after(np.array([173, 173, 173], np.int32), B, code), s_host)
    # a second batch continues the same stream
    rx_h2 = nb.AWGNChannel_CPU(s_host, sigma, code, cw)
    rx_d2 = nb.AWGNChannel_GPU(s_dev, sigma, code, torch.from_numpy(cw).cuda(), 1).cpu().numpy()[0]
    assert np.abs(rx_h2 - rx_d2).max() < 1e-6 and np.array_equal(s_host, s_dev)


@pytest.mark.parametrize("snr", [2, 3, 5])
@pytest.mark.parametrize("layered", [False, True])
def test_tmm_matches_reference_dump(nb, code, orc, snr, layered):
    """Trellis min-max decoders (decoder_method 1 / 3) against the reference's own Decoding_TMM / Decoding_layered_TMM:
    symbols, iteration counts, return flags and the final LLR / L_c2v state of 16 frames per Eb/N0, bit for bit."""
    g = np.load(os.path.join(GOLDEN, "nb_ref_%s_%ddB.npz" % ("ltmm" if layered else "tmm", snr)))
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), float(g["sigma"]))
    r = nb.Decoding_TMM(code, Lch, int(g["maxit"]), layered=layered, want_state=True)
    torch.cuda.synchronize()
    out, it, ok = r["DecodeOutput"].cpu().numpy(), r["iter_number"].cpu().numpy(), r["ok"].cpu().numpy()
    LLR, c2v = r["LLR"].cpu().numpy(), r["L_c2v"].cpu().numpy()
    assert np.array_equal(it, g["it"]) and np.array_equal(ok, g["ok"])
    assert np.array_equal(out, g["out"])
    for fr in range(out.shape[0]):
        assert orc.fold_hash(LLR[fr]) == int(g["LLR_hash"][fr]), "LLR frame %d" % fr
        assert orc.fold_hash(c2v[fr]) == int(g["c2v_hash"][fr]), "c2v frame %d" % fr
    for i, fr in enumerate(g["full_frames"]):
        assert np.array_equal(LLR[fr].view(np.uint32), g["full_LLR"][i].view(np.uint32))
        assert np.array_equal(c2v[fr].view(np.uint32), g["full_c2v"][i].view(np.uint32))


@pytest.mark.parametrize("layered", [False, True])
def test_tmm_ties_and_extremes_vs_oracle(nb, code, ocode, orc, layered):
    """Heavy ties (first-minimum rules, first-best path), zeros, huge and denormal inputs; per-iteration states."""
    rng = np.random.default_rng(5)
    B = 4
    Lch = rng.integers(-3, 4, size=(B, code.N, code.q - 1)).astype(np.float32)
    Lch[0, :, ::5] = -0.0
    Lch[1] *= 1e30
    Lch[2] *= 1e-40
    for maxit in (1, 2, 6):
        r = nb.Decoding_TMM(code, torch.from_numpy(Lch).cuda(), maxit, layered=layered, want_state=True)
        torch.cuda.synchronize()
        for b in range(B):
            want = orc.nb_tmm_decode(ocode, Lch[b], maxit, layered=layered, want_state=True)
            assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
            assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
            assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
            assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))


def test_tmm_large_batch_vs_oracle(nb, code, ocode, orc):
    """256 frames of the reference stream at Eb/N0 = 2.5 dB through both schedules; every frame against the oracle."""
    cw = np.loadtxt(os.path.join(DATA, "nb", "codeword_bds_gf64.txt"), dtype=np.int32)
    seed = np.array([173, 173, 173], np.int32)
    sigma = nb.sigma_of(2.5, code.rate)
    rx = nb.AWGNChannel_GPU(seed, sigma, code, torch.from_numpy(cw).cuda(), 256)
    Lch = nb.Demodulate(code, rx, sigma)
    Lh = Lch.cpu().numpy()
    for layered in (False, True):
        r = nb.Decoding_TMM(code, Lch, 20, layered=layered)
        torch.cuda.synchronize()
        out, it, ok = r["DecodeOutput"].cpu().numpy(), r["iter_number"].cpu().numpy(), r["ok"].cpu().numpy()
        for b in range(0, 256, 3):
            want = orc.nb_tmm_decode(ocode, Lh[b], 20, layered=layered)
            assert it[b] == want["it"] and ok[b] == want["ok"] and np.array_equal(out[b], want["out"]), (layered, b)


def test_log_qspa_is_ems_q_dcm1(nb, code, ocode, orc):
    """decoder_method 2 (Simulation.cpp:63-66): Decoding_EMS(GFQ, maxdc - 1) -- every configuration of every row."""
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    seed = np.array([173, 173, 173], np.int32)
    sigma = nb.sigma_of(3.0, code.rate)
    rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw) for _ in range(2)])
    Lch = nb.Demodulate(code, torch.from_numpy(rx).cuda(), sigma)
    r = nb.Decoding_EMS(code, Lch, code.q, code.dc - 1, 2, want_state=True)
    torch.cuda.synchronize()
    for b in range(2):
        want = orc.nb_ems_decode(ocode, orc.nb_demodulate(ocode, rx[b], sigma), code.q, code.dc - 1, 2, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))


def _synthetic_code(tmp_path, nb, q, poly, N, dv, dc, seed):
    """A random (dv, dc)-regular GF(q) code written in the reference's matrix and GF-table formats (Appendix C), so that the
    product reader, the oracle reader and both decoders see the same files.  The reference ships GF(64) only; this
    exercises the q = 16 / 32 instantiations (lanes beyond q idle) and irregular slot orders."""
    rng = np.random.default_rng(seed)
    M = N * dv // dc
    while True:  # configuration-model bipartite graph without parallel edges
        stubs = np.repeat(np.arange(N), dv)
        rng.shuffle(stubs)
        rows = stubs.reshape(M, dc)
        if all(len(set(r)) == dc for r in rows):
            break
    gf = rng.integers(1, q, size=(M, dc))
    vn_edges = [[] for _ in range(N)]
    for r in range(M):
        for t in range(dc):
            vn_edges[rows[r, t]].append((r, gf[r, t]))
    mpath, gpath = str(tmp_path / ("code%d.txt" % q)), str(tmp_path / ("gf%d.txt" % q))
    with open(mpath, "w") as f:
        f.write("%d %d %d\n%d %d\n" % (N, M, q, dv, dc))
        f.write(" ".join([str(dv)] * N) + "\n" + " ".join([str(dc)] * M) + "\n")
        for i in range(N):
            f.write(" ".join("%d %d" % (r + 1, h) for r, h in vn_edges[i]) + "\n")
        for r in range(M):
            f.write(" ".join("%d %d" % (rows[r, t] + 1, gf[r, t]) for t in range(dc)) + "\n")
    mul, add, inv = nb.GFInitial(q, primitive_poly=poly)
    with open(gpath, "w") as f:
        f.write("GF(%d) with Primitive Polynomial: %d.\nMultiply Table:\n" % (q, poly))
        for a in range(q):
            f.write(" ".join(str(int(x)) for x in mul[a]) + "\n")
        f.write("Add Table:\n")
        for a in range(q):
            f.write(" ".join(str(int(x)) for x in add[a]) + "\n")
        f.write("Inverse Table:\n" + " ".join(str(int(x)) for x in inv) + "\n")
    return mpath, gpath


@pytest.mark.parametrize("q,poly,N", [(16, 19, 48), (32, 37, 60)])
def test_other_fields_synthetic_codes(nb, orc, tmp_path, q, poly, N):
    mpath, gpath = _synthetic_code(tmp_path, nb, q, poly, N, 2, 4, seed=q)
    mul, _, _ = nb.GFInitial(q, gpath)
    code = nb.NBCode(mpath, mul)
    ocode = orc.NBCode(mpath, gpath)
    assert (code.N, code.M, code.q) == (N, N // 2, q)
    rng = np.random.default_rng(q + 1)
    B = 6
    Lch = (rng.standard_normal((B, N, q - 1)) * 2.0).astype(np.float32)
    Lch[0] = np.round(Lch[0])  # ties
    Lt = torch.from_numpy(Lch).cuda()
    r = nb.Decoding_EMS(code, Lt, 2, 2, 4, want_state=True)
    t1 = nb.Decoding_TMM(code, Lt, 4, layered=False, want_state=True)
    t3 = nb.Decoding_TMM(code, Lt, 4, layered=True, want_state=True)
    torch.cuda.synchronize()
    for b in range(B):
        want = orc.nb_ems_decode(ocode, Lch[b], 2, 2, 4, want_state=True)
        assert int(r["iter_number"][b]) == want["it"] and int(r["ok"][b]) == want["ok"]
        assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), want["out"])
        assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
        assert np.array_equal(r["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))
        for got, layered in ((t1, False), (t3, True)):
            want = orc.nb_tmm_decode(ocode, Lch[b], 4, layered=layered, want_state=True)
            assert int(got["iter_number"][b]) == want["it"] and int(got["ok"][b]) == want["ok"], (layered, b)
            assert np.array_equal(got["DecodeOutput"][b].cpu().numpy(), want["out"])
            assert np.array_equal(got["LLR"][b].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
            assert np.array_equal(got["L_c2v"][b].cpu().numpy().view(np.uint32), want["c2v"].view(np.uint32))


@pytest.mark.parametrize("method,tag", [(0, "nb_ref"), (1, "nb_ref_tmm"), (3, "nb_ref_ltmm")])
def test_reference_signature_shim_cpp_harness(nb, code, orc, tmp_path, method, tag):
    """A C++ harness in the reference's calling style (pointer-rich VN[] / CN[], one frame per call) drives
    Decoding_EMS / Decoding_TMM / Decoding_layered_TMM with the reference's signatures (shim/nbldpc_ref_shim.*):
    return flags, iteration counts and the hashes of DecodeOutput, VN[].LLR and CN[].L_c2v equal the reference's own."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from cuda_ldpc_amd._lib import SO_PATH
    root = os.path.dirname(os.path.dirname(SO_PATH))
    exe = str(tmp_path / "nb_ref_style_harness")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "shim"),
                           os.path.join(root, "tests", "cpp", "nb_ref_style_harness.cpp"), os.path.join(root, "shim", "nbldpc_ref_shim.hip"),
                           "-o", exe, "-L", os.path.dirname(SO_PATH), "-lcuda_ldpc_amd", "-Wl,-rpath," + os.path.dirname(SO_PATH), "-pthread"])
    g = np.load(os.path.join(GOLDEN, "%s_3dB.npz" % tag))
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), float(g["sigma"])).cpu().numpy()
    lpath = str(tmp_path / "lch.bin")
    Lch.tofile(lpath)
    frames = Lch.shape[0]
    out = subprocess.check_output([exe, os.path.join(NB, "BDS.576.288.GF.64.txt"), os.path.join(NB, "GF", "Arith.Table.GF.64.txt"), lpath,
                                   str(frames), str(method)]).decode().strip().splitlines()
    assert len(out) == frames, out
    for fr, line in enumerate(out):
        want = "frame %d ok=%d it=%d out=%08x LLR=%08x c2v=%08x" % (fr, int(g["ok"][fr]), int(g["it"][fr]), orc.fold_hash(g["out"][fr]),
                                                                    int(g["LLR_hash"][fr]), int(g["c2v_hash"][fr]))
        assert line == want, (line, want)


@pytest.fixture(scope="module")
def nb_main_exe(tmp_path_factory):
    """tests/cpp/nb_ref_main_style_sweep.cpp + shim/nbldpc_ref_shim.hip linked against the library: built once per module."""
    return _build_nb_main(tmp_path_factory.mktemp("nb_main"))


def _build_nb_main(tmp_path):
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from cuda_ldpc_amd._lib import SO_PATH
    root = os.path.dirname(os.path.dirname(SO_PATH))
    exe = str(tmp_path / "nb_ref_main_style_sweep")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "shim"),
                           os.path.join(root, "tests", "cpp", "nb_ref_main_style_sweep.cpp"), os.path.join(root, "shim", "nbldpc_ref_shim.hip"),
                           "-o", exe, "-L", os.path.dirname(SO_PATH), "-lcuda_ldpc_amd", "-Wl,-rpath," + os.path.dirname(SO_PATH), "-pthread"])
    return exe


def _run_nb_main(exe, *args):
    import subprocess
    out = subprocess.check_output([exe] + [str(a) for a in args], cwd=NB).decode()  # cwd: the reference's relative paths (define.h:23-24, GF.cpp:81)
    assert "task finish" in out, out
    pts = [ln.split() for ln in out.splitlines() if ln.startswith("POINT")]
    frames = [ln.split() for ln in out.splitlines() if ln.startswith("FRAME")]
    rows = [ln for ln in out.splitlines() if ln.startswith(" ") and ln.rstrip().endswith("sec")]
    return pts, frames, rows


@pytest.mark.parametrize("cpu_gpu", [1, 0])
def test_reference_main_style_nb_sweep_cpp_harness(nb, code, ocode, orc, nb_main_exe, cpu_gpu):
    """The NB half of the drop-in boundary: a main() written like myNBLDPC/src/main.cu:14-268 -- Get_H, GFInitial,
    Get_CONSTELLATION, Modulate, the Eb/N0 loop with Simulation_GPU (CPU_GPU 1, the reference's default) or Simulation_CPU, all
    with the reference's signatures -- linked against shim/nbldpc_ref_shim.hip and the library only.  The shim decodes 64 frames
    per launch; per point, frames / error frames / symbol errors / Total_Iteration and the AWGN seeds equal a frame-by-frame
    replay of decode_once_* (Simulation.cpp:115-158) through the oracle, and the first 16 frames at 3 dB equal the
    reference's own dump (tests/golden/nb_ref_3dB.npz)."""
    exe = nb_main_exe
    cwf = os.path.join(NB, "codeword_bds_gf64.txt")
    pts, frames, rows = _run_nb_main(exe, "BDS.576.288.GF.64.txt", cwf, 64, 4, 2, 0, cpu_gpu, 2.0, 3.01, 1.0, 6, 20, 64, 0, 0)
    assert len(pts) == 2 and len(rows) == 2  # 2.0 and 3.0 dB; one result row per finished point (Simulation.cpp:198,241)
    cw = np.loadtxt(cwf, dtype=np.int32)
    for p, snr in zip(pts, (2.0, 3.0)):
        assert abs(float(p[1]) - snr) < 1e-6
        sigma = orc.nb_sigma(snr, ocode.rate)
        oseed = np.array([173, 173, 173], np.int32)
        nfr = errf = errb = its = 0
        while errf < 6 or nfr < 20:  # Simulation.cpp:115
            rx, Lch = orc.nb_channel(ocode, cw, oseed, sigma)
            r = orc.nb_ems_decode(ocode, Lch, 2, 2, 20)
            nfr += 1
            its += r["it"]
            e = int((r["out"] != cw).sum())
            errb += e
            errf += 1 if e else 0
        assert [int(x) for x in p[2:9]] == [nfr, errf, errb, its] + oseed.tolist(), (p, nfr, errf, errb, its, oseed)
    # the first 16 frames of the 3 dB point against the reference's own outputs
    g = np.load(os.path.join(GOLDEN, "nb_ref_3dB.npz"))
    pts, frames, _ = _run_nb_main(exe, "BDS.576.288.GF.64.txt", cwf, 64, 4, 2, 0, cpu_gpu, 3.0, 3.01, 1.0, 50, 1000, 512, 0, 16)
    assert len(frames) == 16
    for fr, ln in enumerate(frames):
        assert [int(ln[1]), int(ln[2]), int(ln[3])] == [fr, int(g["ok"][fr]), int(g["it"][fr])], ln
        assert int(ln[4], 16) == orc.fold_hash(g["out"][fr]), ln
    assert int(pts[0][2]) >= 1000 and int(pts[0][3]) >= 50  # the reference's stop rule (define.h:52-53)


@pytest.mark.parametrize("method,cpu_gpu,least_err,least_frames", [(1, 1, 30, 1500), (3, 0, 30, 1500), (2, 1, 3, 48)])  # log-QSPA walks conf(64, 3): few frames
def test_nb_sweep_cpp_harness_fast_mode_equals_python_loop(nb, code, nb_main_exe, method, cpu_gpu, least_err, least_frames):
    """The same harness with the device-side channel (cfg.device_channel = 1: no host loop, no upload) and the other
    decoder_method values: counters and seeds equal cuda_ldpc_amd.nb_simulation.Simulation_GPU on the same device channel
    (same draws, same device libm), whatever the two batch sizes are."""
    from cuda_ldpc_amd.nb_simulation import NBSim, Simulation_GPU
    exe = nb_main_exe
    cwf = os.path.join(NB, "codeword_bds_gf64.txt")
    cw = np.loadtxt(cwf, dtype=np.int32)
    pts, _, _ = _run_nb_main(exe, "BDS.576.288.GF.64.txt", cwf, 64, 4, 2, method, cpu_gpu, 2.5, 2.51, 1.0, least_err, least_frames, 1000, 1, 0)
    sigma = nb.sigma_of(2.5, code.rate)
    seed = np.array([173, 173, 173], np.int32)
    SIM = NBSim(2.5)
    assert Simulation_GPU(code, seed, sigma, SIM, cw, batch=384 if method != 2 else 32, leastErrorFrames=least_err, leastTestFrames=least_frames, device_channel=True,
                          decoder_method=method) == 1
    assert [int(x) for x in pts[0][2:9]] == [SIM.num_Frames, SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration] + seed.tolist()


def test_nb_simulation_gpu_refuses_layered_tmm_like_the_reference(nb, nb_main_exe):
    """decode_once_gpu prints "unfinished" and exits for decoder_method 3 (Simulation.cpp:140-144); so does the shim's Simulation_GPU."""
    import subprocess
    exe = nb_main_exe
    r = subprocess.run([exe, "BDS.576.288.GF.64.txt", os.path.join(NB, "codeword_bds_gf64.txt"), "64", "4", "2", "3", "1", "3.0", "3.01", "1.0", "5", "10",
                        "64", "0", "0"], cwd=NB, capture_output=True, text=True)
    assert r.returncode == 0 and "unfinished" in r.stdout and "POINT" not in r.stdout


# ---- QAM constellations (n_QAM = q).  PARITY UNPINNED against the reference (define.h fixes n_QAM = 2; nothing in its tree
# records an output of these branches): the checker is the oracle's restatement of the source text. -----------------------------
@pytest.fixture(scope="module")
def qam64(nb):
    return nb.Get_CONSTELLATION(os.path.join(NB, "Constellation", "GRAY_64QAM.txt"), 64)


def test_qam_demodulate_and_decode_vs_oracle(nb, code, ocode, orc, qam64):
    """Demodulate (n_QAM != 2 branch) on the device equals the restatement bit for bit; the decoders then see the same
    L_ch and agree with the oracle as for BPSK (EMS and trellis min-max), frames that converge and frames that do not."""
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    con_dev = torch.from_numpy(qam64).cuda()
    for snr, B in ((11.0, 6), (14.0, 6)):  # the waterfall of this code with 64-QAM and EMS(2,2) is near Eb/N0 = 11-12 dB
        sigma = nb.sigma_of(snr, code.rate, 0, 64)
        seed = np.array([173, 173, 173], np.int32)
        rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw, CONSTELLATION=qam64) for _ in range(B)])
        Lch = nb.Demodulate(code, torch.from_numpy(rx).cuda(), sigma, CONSTELLATION=con_dev)
        want_L = np.stack([orc.nb_demodulate_qam(ocode, rx[b], sigma, qam64) for b in range(B)])
        assert np.array_equal(Lch.cpu().numpy().view(np.uint32), want_L.view(np.uint32))
        r = nb.Decoding_EMS(code, Lch, 2, 2, 20, want_state=True)
        t = nb.Decoding_TMM(code, Lch, 20, layered=False)
        torch.cuda.synchronize()
        for b in range(B):
            w = orc.nb_ems_decode(ocode, want_L[b], 2, 2, 20, want_state=True)
            assert int(r["iter_number"][b]) == w["it"] and int(r["ok"][b]) == w["ok"]
            assert np.array_equal(r["DecodeOutput"][b].cpu().numpy(), w["out"])
            assert np.array_equal(r["LLR"][b].cpu().numpy().view(np.uint32), w["LLR"].view(np.uint32))
            wt = orc.nb_tmm_decode(ocode, want_L[b], 20, layered=False)
            assert int(t["iter_number"][b]) == wt["it"] and np.array_equal(t["DecodeOutput"][b].cpu().numpy(), wt["out"])
    assert all(int(x) == 1 for x in r["ok"]) and np.array_equal(r["DecodeOutput"].cpu().numpy(), np.tile(cw, (B, 1)))  # 14 dB: all decode


@pytest.mark.parametrize("snr", [11, 14])
def test_qam64_matches_reference_dump(nb, code, orc, qam64, snr):
    """64-QAM through the reference itself (built with n_QAM 64, tests/golden/nb_ref_qam64_*.npz): the host channel reproduces its
    complex samples, the device Demodulate its L_ch bits, the decoder its symbols / iteration counts / LLR and c2v hashes."""
    g = np.load(os.path.join(GOLDEN, "nb_ref_qam64_%ddB.npz" % snr))
    sigma = float(g["sigma"])
    seed = np.array([173, 173, 173], np.int32)
    rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, g["cw"], CONSTELLATION=qam64) for _ in range(g["rx"].shape[0])])
    assert np.array_equal(rx.view(np.uint32), g["rx"].view(np.uint32))
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), sigma, CONSTELLATION=torch.from_numpy(qam64).cuda())
    assert np.array_equal(Lch.cpu().numpy().view(np.uint32), g["Lch"].view(np.uint32))
    r = nb.Decoding_EMS(code, Lch, 2, 2, int(g["maxit"]), want_state=True)
    torch.cuda.synchronize()
    assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["ok"].cpu().numpy(), g["ok"])
    assert np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"])
    for fr in range(g["rx"].shape[0]):
        assert orc.fold_hash(r["LLR"][fr].cpu().numpy()) == int(g["LLR_hash"][fr])
        assert orc.fold_hash(r["L_c2v"][fr].cpu().numpy()) == int(g["c2v_hash"][fr])


def test_qam_device_channel_and_simulation_loop(nb, code, ocode, orc, qam64):
    """Device-side QAM channel: same draws as the host loop (seeds equal, samples equal up to the device libm); the simulation
    loop with a constellation stops where a per-frame replay with the oracle stops."""
    from cuda_ldpc_amd.nb_simulation import NBSim, Simulation_GPU
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    con_dev = torch.from_numpy(qam64).cuda()
    sigma = nb.sigma_of(12.0, code.rate, 0, 64)
    s_host, s_dev = np.array([173, 173, 173], np.int32), np.array([173, 173, 173], np.int32)
    B = 11
    rx_h = np.stack([nb.AWGNChannel_CPU(s_host, sigma, code, cw, CONSTELLATION=qam64) for _ in range(B)])
    rx_d = nb.AWGNChannel_GPU(s_dev, sigma, code, torch.from_numpy(cw).cuda(), B, CONSTELLATION=con_dev).cpu().numpy()
    assert np.array_equal(s_host, s_dev) and rx_d.shape == (B, 96, 2)
    same = (rx_h.view(np.int32) == rx_d.view(np.int32)).mean()
    assert same > 0.8 and np.abs(rx_h - rx_d).max() < 1e-6, (same, np.abs(rx_h - rx_d).max())
    assert np.array_equal(nb.seed_after(np.array([173, 173, 173], np.int32), B, code, qam=True), s_host)
    snr = 11.5
    sigma = nb.sigma_of(snr, code.rate, 0, 64)
    seed = np.array([173, 173, 173], np.int32)
    SIM = NBSim(snr)
    stop = Simulation_GPU(code, seed, sigma, SIM, cw, batch=64, leastErrorFrames=6, leastTestFrames=20, CONSTELLATION=qam64)
    assert stop == 1
    oseed = np.array([173, 173, 173], np.int32)
    frames = errf = errb = its = 0
    while errf < 6 or frames < 20:
        _, Lch = orc.nb_channel_qam(ocode, cw, oseed, sigma, qam64)
        r = orc.nb_ems_decode(ocode, Lch, 2, 2, 20)
        frames += 1
        its += r["it"]
        e = int((r["out"] != cw).sum())
        errb += e
        errf += 1 if e else 0
    assert (SIM.num_Frames, SIM.num_Error_Frames, SIM.num_Error_Bits, SIM.Total_Iteration) == (frames, errf, errb, its)
    assert np.array_equal(seed, oseed)


@pytest.mark.parametrize("which", ["tmm", "ltmm", "ems64", "gf256"])
def test_persistent_nb_kernels_equal_one_workgroup_per_frame(nb, orc, monkeypatch, which):
    """Batches larger than the resident grid take the persistent form of the GF(q) kernels (frames from an atomic counter): same
    symbols, iteration counts, flags and final LLR / L_c2v bits as a code object created under NBLDPC_NO_PERSIST=1 (one workgroup
    per frame; the switch is read when the code is created), plus a sample of frames against the oracle."""
    q = 256 if which == "gf256" else 64
    mfile = "LDPC_N96_K48_GF256_d1_exp.txt" if q == 256 else "BDS.576.288.GF.64.txt"
    gff = os.path.join(NB, "GF", "Arith.Table.GF.%d.txt" % q)
    mul, _, _ = nb.GFInitial(q, gff)
    code = nb.NBCode(os.path.join(NB, mfile), mul)
    monkeypatch.setenv("NBLDPC_NO_PERSIST", "1")
    plain = nb.NBCode(os.path.join(NB, mfile), mul)
    monkeypatch.delenv("NBLDPC_NO_PERSIST")
    ocode = orc.NBCode(os.path.join(NB, mfile), gff)
    B = 1536
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32) if q == 64 else np.zeros(code.N, np.int32)
    seed = np.array([173, 173, 173], np.int32)
    sigma = nb.sigma_of(4.0 if q == 256 else 2.8, code.rate)
    rx = nb.AWGNChannel_GPU(seed, sigma, code, torch.from_numpy(cw).cuda(), B)
    Lch = nb.Demodulate(code, rx, sigma)

    def run(c):
        if which in ("tmm", "ltmm"):
            return nb.Decoding_TMM(c, Lch, 20, layered=(which == "ltmm"), want_state=True)
        return nb.Decoding_EMS(c, Lch, 2, 2, 20, want_state=True)
    a, b = run(code), run(plain)
    torch.cuda.synchronize()
    for k in ("DecodeOutput", "iter_number", "ok"):
        assert torch.equal(a[k], b[k]), k
    for k in ("LLR", "L_c2v"):
        assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), k
    assert len(set(a["iter_number"].cpu().tolist())) > 3
    Lh = Lch.cpu().numpy()
    for f in (0, 1, B // 2 + 3, B - 1):
        if which in ("tmm", "ltmm"):
            w = orc.nb_tmm_decode(ocode, Lh[f], 20, layered=(which == "ltmm"), want_state=True)
        else:
            w = orc.nb_ems_decode(ocode, Lh[f], 2, 2, 20, want_state=True)
        assert int(a["iter_number"][f]) == w["it"] and int(a["ok"][f]) == w["ok"] and np.array_equal(a["DecodeOutput"][f].cpu().numpy(), w["out"])
        assert np.array_equal(a["LLR"][f].cpu().numpy().view(np.uint32), w["LLR"].view(np.uint32))


# ---- k_nb_ems2: two frames in flight per workgroup (nbldpc_pipe_kernel.hpp); every call without the L_c2v output takes it ----------
@pytest.mark.parametrize("snr", [2, 3, 5])
def test_pipeline_kernel_matches_reference_dump(nb, code, orc, snr):
    """The reference's own channel samples and outputs (Decoding_EMS of the reference, tests/golden/nb_ref_*.npz) through the
    two-frame pipeline kernel: symbols, iteration counts, return flags and the bits of the final LLR of all 16 frames."""
    g = np.load(os.path.join(GOLDEN, "nb_ref_%ddB.npz" % snr))
    Lch = nb.Demodulate(code, torch.from_numpy(g["rx"]).cuda(), float(g["sigma"]))
    r = nb.Decoding_EMS(code, Lch, 2, 2, int(g["maxit"]), want_state="llr")
    torch.cuda.synchronize()
    assert r["L_c2v"] is None
    assert np.array_equal(r["iter_number"].cpu().numpy(), g["it"]) and np.array_equal(r["ok"].cpu().numpy(), g["ok"])
    assert np.array_equal(r["DecodeOutput"].cpu().numpy(), g["out"])
    LLR = r["LLR"].cpu().numpy()
    for fr in range(LLR.shape[0]):
        assert orc.fold_hash(LLR[fr]) == int(g["LLR_hash"][fr]), "LLR frame %d" % fr


@pytest.mark.parametrize("B,snr,maxit", [(3001, 2.8, 20), (2, 3.0, 20), (3, 1.0, 5), (700, 2.0, 20), (513, 6.0, 20), (640, 3.2, 1)])
def test_pipeline_kernel_equals_the_one_frame_kernel(nb, code, monkeypatch, B, snr, maxit):
    """k_nb_ems2 against k_nb_ems (a code object created under NBLDPC_NO_PIPE=1) on whole batches: symbols, iteration counts,
    flags and final LLR bits -- odd batches, two frames, frames that all fail, frames that all pass at once, maxIT 1."""
    mul, _, _ = nb.GFInitial(64, os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    monkeypatch.setenv("NBLDPC_NO_PIPE", "1")
    plain = nb.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), mul)
    monkeypatch.delenv("NBLDPC_NO_PIPE")
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    seed = np.array([173, 173, 173], np.int32)
    sigma = nb.sigma_of(snr, code.rate)
    Lch = nb.Demodulate(code, nb.AWGNChannel_GPU(seed, sigma, code, torch.from_numpy(cw).cuda(), B), sigma)
    if B == 700:
        Lch = torch.round(Lch)  # ties everywhere: the sort's repeat path, first-maximum decisions
    a = nb.Decoding_EMS(code, Lch, 2, 2, maxit, want_state="llr")
    b = nb.Decoding_EMS(plain, Lch, 2, 2, maxit, want_state="llr")
    torch.cuda.synchronize()
    for k in ("DecodeOutput", "iter_number", "ok"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["LLR"].view(torch.int32), b["LLR"].view(torch.int32))
    its = a["iter_number"].cpu().numpy()
    if B == 3001:
        assert len(set(its.tolist())) > 5
    c = nb.Decoding_EMS(code, Lch, 2, 2, maxit)  # no state outputs at all: the production call
    torch.cuda.synchronize()
    for k in ("DecodeOutput", "iter_number", "ok"):
        assert torch.equal(a[k], c[k]), k


def _irregular_gf64_code(tmp_path, nb, N, dc, drop, seed):
    """A random GF(64) code with column weights 1 / 2 and row weights dc - 1 / dc: a (2, dc)-regular graph with `drop` edges taken
    out (no column left without an edge, every row keeps at least two), in the reference's matrix format."""
    rng = np.random.default_rng(seed)
    M = N * 2 // dc
    while True:
        stubs = np.repeat(np.arange(N), 2)
        rng.shuffle(stubs)
        rows = stubs.reshape(M, dc)
        if all(len(set(r)) == dc for r in rows):
            break
    gf = rng.integers(1, 64, size=(M, dc))
    edges = [(r, int(rows[r, t]), int(gf[r, t])) for r in range(M) for t in range(dc)]
    colw = np.full(N, 2)
    roww = np.full(M, dc)
    for k in rng.permutation(len(edges)):
        if drop == 0:
            break
        r, c, _ = edges[k]
        if colw[c] == 2 and roww[r] == dc:  # at most one edge out of any row or column
            colw[c] -= 1
            roww[r] -= 1
            edges[k] = None
            drop -= 1
    edges = [e for e in edges if e is not None]
    mpath = str(tmp_path / ("irr_%d_%d.txt" % (N, dc)))
    with open(mpath, "w") as f:
        f.write("%d %d 64\n2 %d\n" % (N, M, dc))
        f.write(" ".join(str(int(w)) for w in colw) + "\n" + " ".join(str(int(w)) for w in roww) + "\n")
        for c in range(N):
            f.write(" ".join("%d %d" % (r + 1, h) for r, cc, h in edges if cc == c) + "\n")
        for r in range(M):
            f.write(" ".join("%d %d" % (c + 1, h) for rr, c, h in edges if rr == r) + "\n")
    return mpath


@pytest.mark.gpu
@pytest.mark.parametrize("N,dc,drop", [(63, 3, 0), (60, 5, 7), (90, 6, 11), (92, 4, 9)])
def test_pipeline_kernel_on_other_gf64_graphs(nb, orc, tmp_path, monkeypatch, N, dc, drop):
    """k_nb_ems2 is offered to every GF(64) code with column weights <= 2 whose columns fit its sorting waves, not only to the
    reference's BDS code: an odd number of columns (half a group of sorts), rows of weight 5 and 6 (two syndrome table reads per
    row, two or three walking waves), columns of weight 1 and rows one short (absent edges inside a group).  Against k_nb_ems on the
    whole batch and against the oracle on its first frames."""
    gpath = os.path.join(NB, "GF", "Arith.Table.GF.64.txt")
    mpath = _irregular_gf64_code(tmp_path, nb, N, dc, drop, seed=N + dc)
    mul, _, _ = nb.GFInitial(64, gpath)
    code = nb.NBCode(mpath, mul)
    monkeypatch.setenv("NBLDPC_NO_PIPE", "1")
    plain = nb.NBCode(mpath, mul)
    monkeypatch.delenv("NBLDPC_NO_PIPE")
    ocode = orc.NBCode(mpath, gpath)
    rng = np.random.default_rng(N * dc)
    B = 37
    Lch = (rng.standard_normal((B, N, 63)) * 3.0).astype(np.float32)
    Lch[1] = np.round(Lch[1])  # ties
    Lch[2:12] += 6.0 * (rng.random((10, N, 63)) < 0.02)  # a few strong symbols: frames that converge at different iterations
    Lt = torch.from_numpy(Lch).cuda()
    a = nb.Decoding_EMS(code, Lt, 2, 2, 6, want_state="llr")
    assert code.last_kernel == "k_nb_ems2"
    b = nb.Decoding_EMS(plain, Lt, 2, 2, 6, want_state="llr")
    assert plain.last_kernel == "k_nb_ems"
    torch.cuda.synchronize()
    for k in ("DecodeOutput", "iter_number", "ok"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["LLR"].view(torch.int32), b["LLR"].view(torch.int32))
    for f in range(3):
        want = orc.nb_ems_decode(ocode, Lch[f], 2, 2, 6, want_state=True)
        assert int(a["iter_number"][f]) == want["it"] and int(a["ok"][f]) == want["ok"]
        assert np.array_equal(a["DecodeOutput"][f].cpu().numpy(), want["out"])
        assert np.array_equal(a["LLR"][f].cpu().numpy().view(np.uint32), want["LLR"].view(np.uint32))
