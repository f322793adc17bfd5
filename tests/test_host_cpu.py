"""CPU-only tests: the C-ABI library loads and exports every declared symbol; host logic (graph builders,
readers, GF tables, channel generators) equals the oracle's restatement of the reference."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import DATA, ROOT

BL = os.path.join(DATA, "bldpc")
NB = os.path.join(DATA, "nb")


@pytest.fixture(scope="module")
def C():
    import cuda_ldpc_amd
    return cuda_ldpc_amd


@pytest.fixture(scope="module")
def nbm(C):
    from cuda_ldpc_amd import nbldpc
    return nbldpc


def test_abi_exports_every_declared_symbol(C):
    from cuda_ldpc_amd._lib import SO_PATH
    so = ctypes.CDLL(SO_PATH)
    declared = []
    for h in ("bldpc.h", "nbldpc.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared += re.findall(r"\b((?:nb)?bldpc_\w+)\s*\(", text) + re.findall(r"\b(nbldpc_\w+)\s*\(", text)
    declared = sorted(set(declared))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(so, name), "missing export %s" % name


def test_no_oracle_on_product_path():
    """The product package must not import / link / dlopen the oracle."""
    pkg = os.path.join(ROOT, "cuda_ldpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read().lower()
                assert "oracle" not in src and "pyoracle" not in src and "orc_" not in src, f
    so = open(os.path.join(pkg, "libcuda_ldpc_amd.so"), "rb").read()
    assert b"liboracle" not in so and b"orc_bldpc" not in so and b"orc_nb" not in so


ALL_MATRICES = sorted(f for f in os.listdir(BL) if f.endswith(".txt"))


@pytest.mark.parametrize("fn", ALL_MATRICES)
def test_get_h_transform_h_all_matrices(C, orc, fn):
    m = re.match(r"J(\d+)_L(\d+)_Z(\d+)", fn)
    J, L, Z = (int(x) for x in m.groups()) if m else (12, 69, 256)
    H, wc, wv = C.Get_H(os.path.join(BL, fn), J, L)
    for lit in (False, True):
        oc = orc.BinaryCode(os.path.join(BL, fn), J, L, Z, literal=lit)
        assert np.array_equal(H, oc.H) and np.array_equal(wc, oc.wc) and np.array_equal(wv, oc.wv)
        assert np.array_equal(C.Transform_H(H, J, L, Z, wc, wv, as_written=lit), oc.addr)
    # the corrected table is a bijection between edges and Memory_RQ slots; the as-written one is not
    addr = C.Transform_H(H, J, L, Z, wc, wv).reshape(L * Z, -1)
    used = np.concatenate([addr[l * Z:(l + 1) * Z, :wv[l]].reshape(-1) for l in range(L)])
    assert len(np.unique(used)) == used.size == int((H != -1).sum()) * Z
    lit = C.Transform_H(H, J, L, Z, wc, wv, as_written=True).reshape(L * Z, -1)
    usedl = np.concatenate([lit[l * Z:(l + 1) * Z, :wv[l]].reshape(-1) for l in range(L)])
    if fn == "J4_L24_Z96_BlockH.txt":
        assert usedl.size - len(np.unique(usedl)) == 1542  # collisions counted in SURVEY F3


def test_binary_channel_equals_oracle(C, orc):
    for snr in (0.0, 3.0, 12.999992):
        s1 = np.array([173, 173, 173], np.int32)
        s2 = s1.copy()
        a = C.AWGNChannel_CPU(s1, C.sigma_of(snr), 2304, 5)
        b = orc.bldpc_awgn(s2, orc.bldpc_sigma(snr), 2304, 5)
        assert np.array_equal(a.reshape(-1).view(np.uint32), b.view(np.uint32)) and np.array_equal(s1, s2)
    assert C.sigma_of(2.0, snrtype=0, rate=0.5) == orc.bldpc_sigma(2.0, 0, 0.5)
    cw = np.random.default_rng(0).integers(0, 2, (100, 3)).astype(np.int32)
    s1 = np.array([1, 2, 3], np.int32); s2 = s1.copy()
    assert np.array_equal(C.AWGNChannel_CPU(s1, 0.7, 100, 3, CodeWord=cw).reshape(-1), orc.bldpc_awgn(s2, 0.7, 100, 3, codeword=cw))


def test_host_errors(C):
    with pytest.raises(Exception):
        C.Get_H(os.path.join(BL, "J4_L24_Z96_BlockH.txt"), 40, 24)   # file too short
    with pytest.raises(Exception):
        C.Transform_H(np.full(96, 200, np.int32), 4, 24, 96, np.zeros(5, np.int32) + 20, np.zeros(25, np.int32) + 4)


def test_gf_tables(nbm, orc):
    oc = orc.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    mul, add, inv = nbm.GFInitial(64, os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    assert np.array_equal(mul.reshape(-1), oc.mul) and np.array_equal(add.reshape(-1), oc.add) and np.array_equal(inv, oc.inv)
    gmul, gadd, ginv = nbm.GFInitial(64, primitive_poly=67)  # title line of the reference's table file
    assert np.array_equal(gmul, mul) and np.array_equal(gadd, add) and np.array_equal(ginv, inv)
    with pytest.raises(Exception):
        nbm.GFInitial(64, primitive_poly=64 + 1)  # x^6+1 is reducible


@pytest.mark.skipif(not os.path.isdir("/root/reference/myNBLDPC/GF"), reason="reference tree absent (GPU box)")
@pytest.mark.parametrize("q,poly", [(4, 7), (8, 11), (16, 19), (32, 37), (64, 67), (128, 137), (256, 285)])
def test_gf_generator_matches_every_reference_table(nbm, q, poly):
    path = "/root/reference/myNBLDPC/GF/Arith.Table.GF.%d.txt" % q
    title = open(path).readline()
    assert str(poly) in title, title
    mul, add, inv = nbm.GFInitial(q, path)
    gmul, gadd, ginv = nbm.GFInitial(q, primitive_poly=poly)
    assert np.array_equal(gmul, mul) and np.array_equal(gadd, add) and np.array_equal(ginv, inv)


def test_nb_channel_equals_oracle_and_reference(nbm, orc):
    g = np.load(os.path.join(ROOT, "tests", "golden", "nb_ref_3dB.npz"))
    oc = orc.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))

    class Shape:  # host-only stand-in for NBCode (creating a device code needs a GPU)
        N, m = 96, 6
    assert np.float32(nbm.sigma_of(3.0, oc.rate)) == np.float32(g["sigma"])
    seed = np.array([173, 173, 173], np.int32)
    for fr in range(4):
        rx = nbm.AWGNChannel_CPU(seed, float(g["sigma"]), Shape, g["cw"])
        assert np.array_equal(rx.view(np.uint32), g["rx"][fr].view(np.uint32))  # the reference's own samples


def test_nb_matrix_reader(nbm, orc):
    from cuda_ldpc_amd._lib import lib
    oc = orc.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    dims = np.zeros(5, np.int32)
    p = os.path.join(NB, "BDS.576.288.GF.64.txt").encode()
    assert lib.nbldpc_read_matrix(p, dims.ctypes.data_as(ctypes.c_void_p), None, None, None, None, None, None) == 0
    assert list(dims) == [96, 48, 64, 2, 4]
    arrs = [np.zeros(n, np.int32) for n in (96, 192, 192, 48, 192, 192)]
    assert lib.nbldpc_read_matrix(p, dims.ctypes.data_as(ctypes.c_void_p), *[a.ctypes.data_as(ctypes.c_void_p) for a in arrs]) == 0
    for a, b in zip(arrs, (oc.vn_w, oc.vn_cn, oc.vn_gf, oc.cn_w, oc.cn_vn, oc.cn_gf)):
        assert np.array_equal(a, b)
    assert lib.nbldpc_read_matrix(b"/nonexistent", dims.ctypes.data_as(ctypes.c_void_p), None, None, None, None, None, None) != 0


def test_c_program_links_against_the_abi(C, orc, tmp_path):
    """A plain C translation unit includes include/*.h, links libcuda_ldpc_amd.so and runs the host-side entry points."""
    import subprocess
    from cuda_ldpc_amd._lib import SO_PATH
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(SO_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-o", exe, "-L", libdir, "-lcuda_ldpc_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe, os.path.join(BL, "J4_L24_Z96_BlockH.txt")]).decode()
    oc = orc.BinaryCode(os.path.join(BL, "J4_L24_Z96_BlockH.txt"), 4, 24, 96)
    assert "Wc=20 Wv=4" in out and ("addrsum=%d" % int(oc.addr.astype(np.int64).sum())) in out
    assert "y0=0.865149975" in out            # SURVEY Appendix D.3 anchor (Es/N0 3 dB)
    assert "mul[2][33]=" in out and "can not open file" in out


def test_nb_matrix_with_out_of_range_coefficient_is_rejected(nbm, tmp_path):
    """Coefficients outside GF(q) are refused before any device allocation (no GPU needed).  Zero coefficients are NOT: the
    reference reads its exponent-format files as field elements and decodes with the zeros in place
    (tests/test_nbldpc_gpu.py::test_exponent_format_matrix_matches_reference_dump)."""
    mul, _, _ = nbm.GFInitial(64, os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
    src = open(os.path.join(NB, "BDS.576.288.GF.64.txt")).read().split("\n")
    bad = tmp_path / "bad.txt"
    toks = src[4].split()
    toks[1] = "64"  # first edge of the first variable node: coefficient 64 is not an element of GF(64)
    src[4] = " ".join(toks)
    bad.write_text("\n".join(src))
    with pytest.raises(Exception, match="field element|coefficient"):
        nbm.NBCode(str(bad), mul)


def test_division_shortcut_exhaustive(tmp_path):
    """The EMS kernel's c2v = (float)((double)x / 1.2) (LDPC_Decoder.cpp:309) is computed with one reciprocal
    multiplication and Markstein's correction step (nbldpc_kernel.hpp, nb_div12): checked against the division for
    EVERY finite non-zero float on the host (same IEEE double arithmetic, fma from libm)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "div12")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "div12_exhaustive.c")
    subprocess.check_call([gcc, "-O2", "-fopenmp", "-ffp-contract=off", src, "-lm", "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert "checked 4278190078 floats, 0 mismatches" in out, out


def test_explicit_stack_walk_equals_the_recursion(tmp_path):
    """k_nb_ems_hbm walks a check row's configurations with an explicit stack (nb_hbm_conf, any row weight / Nm / Nc); on the host
    (the function is __host__ __device__) it must leave the same bits in the max array as a plain recursion shaped like the
    reference's ConstructConf (LDPC_Decoder.cpp:319-359): row weights 1 ... 21, q up to 256 (k reaches 256: not a byte)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "hbm_conf")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "hbm_conf_host_test.hip")
    subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", src, "-o", exe], cwd=str(tmp_path))
    out = subprocess.check_output([exe], timeout=120).decode()
    assert out.startswith("OK "), out


def test_local_edge_matching_is_valid(tmp_path):
    """The fused row / half-row kernels keep one block per column out of LDS ("local edges"): the host hands every block column to
    one block row containing it, L / J per row (qc2_local_assign, csrc/bldpc_qc_assign.hpp -- no HIP in that header).  On 1 400 random
    block patterns that admit such an assignment it must find a valid one (also where single-block columns leave no choice), and
    it must refuse rows lighter than their share and L not a multiple of J."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "qla")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "qc_local_assign_host_test.cpp")
    subprocess.check_call([gxx, "-O2", "-std=c++17", src, "-o", exe], cwd=str(tmp_path))
    out = subprocess.check_output([exe], timeout=120).decode()
    assert out.startswith("OK 1400 "), out


def test_qam_constellation_and_channel_equal_the_restatement(nbm, orc):
    """n_QAM != 2 branches (PARITY UNPINNED against the reference: its define.h fixes n_QAM = 2 and its tree holds no output of
    these branches): the library's reader and host channel equal the oracle's restatement of Get_CONSTELLATION / Modulate /
    AWGNChannel_CPU bit for bit, seeds included; the shipped 64-QAM file is a unit-energy Gray-labelled square grid."""
    path = os.path.join(NB, "Constellation", "GRAY_64QAM.txt")
    con = nbm.Get_CONSTELLATION(path, 64)
    assert np.array_equal(con, orc.nb_read_constellation(path, 64))
    assert len({(float(a), float(b)) for a, b in con}) == 64 and abs(float((con.astype(np.float64) ** 2).sum(1).mean()) - 1.0) < 1e-6
    lv = np.unique(np.round(con[:, 0], 6))
    assert len(lv) == 8 and np.allclose(np.diff(lv), lv[1] - lv[0], atol=1e-6)
    step = float(lv[1] - lv[0])
    for a in range(64):  # Gray labelling: nearest neighbours differ in exactly one bit
        for b in range(64):
            if a < b and abs(float(np.hypot(*(con[a] - con[b]))) - step) < 1e-4:
                assert bin(a ^ b).count("1") == 1

    class Shape:
        N, m, q = 96, 6, 64
    cw = np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)
    sigma = nbm.sigma_of(9.0, 0.5, 0, 64)
    assert np.float32(sigma) == np.float32(orc.nb_sigma(9.0, 0.5, 0, 64))
    s1, s2 = np.array([173, 173, 173], np.int32), np.array([173, 173, 173], np.int32)
    for _ in range(3):
        rx = nbm.AWGNChannel_CPU(s1, sigma, Shape, cw, CONSTELLATION=con)
        want, _ = orc.nb_channel_qam(Shape, cw, s2, sigma, con)
        assert rx.shape == (96, 2) and np.array_equal(rx.view(np.uint32), want.view(np.uint32)) and np.array_equal(s1, s2)
    with pytest.raises(Exception):
        nbm.Get_CONSTELLATION(os.path.join(NB, "Constellation", "BPSK.txt"), 64)  # two records only


def test_lcg_quotient_shortcut_exhaustive():
    """The device channel generators form RandomModule's s / m (float division, LDPC_Encoder.cu:51-53) as the double product
    s * (1/m) rounded to float: identical for every state of the three generators."""
    for m in (61967, 63443, 63599):
        x = np.arange(m, dtype=np.int32)
        want = x.astype(np.float32) / np.float32(m)
        got = (x.astype(np.float64) * np.float64(1.0 / m)).astype(np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_nb_reference_signature_shim_host_side(nbm, orc, tmp_path):
    """The NB shim (shim/nbldpc_ref_shim.hpp) without a GPU: the classes have the layout of the reference's include/struct.h:9-71 on
    this ABI (LP64), and the reference-signature host functions -- Get_H, GFInitial + tables, Get_CONSTELLATION, BitToSym, Modulate,
    AWGNChannel_CPU, RandomModule, index_in_VN / index_in_CN, Statistic -- agree with the oracle on the reference's own data files.
    Also proves that the shim and both main()-style sweeps compile and link against the library alone."""
    import shutil
    import subprocess
    from cuda_ldpc_amd._lib import SO_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    libdir = os.path.dirname(SO_PATH)
    common = [hipcc, "-O1", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "shim")]
    link = ["-L", libdir, "-lcuda_ldpc_amd", "-Wl,-rpath," + libdir, "-pthread"]
    exe = str(tmp_path / "nb_shim_host_test")
    subprocess.check_call(common + [os.path.join(ROOT, "tests", "cpp", "nb_shim_host_test.cpp"), os.path.join(ROOT, "shim", "nbldpc_ref_shim.hip"), "-o", exe] + link)
    for src, shim in (("nb_ref_main_style_sweep.cpp", "nbldpc_ref_shim.hip"), ("ref_main_style_sweep.cpp", "ldpc_ref_shim.hip")):
        subprocess.check_call(common + [os.path.join(ROOT, "tests", "cpp", src), os.path.join(ROOT, "shim", shim), "-o", str(tmp_path / src[:-4])] + link)
    nbd = os.path.join(ROOT, "data", "nb")
    ocode = orc.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
    sigma = orc.nb_sigma(3.0, ocode.rate)
    out = subprocess.check_output([exe, "codeword_bds_gf64.txt", "%.9g" % sigma], cwd=nbd).decode().splitlines()
    line = {ln.split()[0]: ln.split()[1:] for ln in out if ln and ln.split()[0].isupper()}
    # include/struct.h:9-71 on LP64: CComplex {float,float}; LDPCCode 6 int + float + ... = 32; VN 2 ptr,int,4 ptr = 56; CN = 32; AWGNChannel 16;
    # Simulation: float SNR @0, double sumTime @8, six longs @16..56, five floats @64..80 -> 88
    assert line["LAYOUT"] == ["CComplex", "8", "LDPCCode", "32", "VN", "56", "CN", "32", "AWGNChannel", "16", "Simulation", "88"]
    assert line["OFFSETS"][:14] == ["Simulation", "0", "8", "16", "24", "32", "40", "48", "56", "64", "68", "72", "76", "80"]
    assert line["OFFSETS2"] == ["VN", "0", "8", "16", "24", "32", "40", "48", "CN", "0", "8", "16", "24", "LDPCCode", "0", "4", "8", "12", "16", "20", "24", "28"]
    cw = np.loadtxt(os.path.join(nbd, "codeword_bds_gf64.txt"), dtype=np.int32)
    g = line["GET_H"]
    assert [int(x) for x in g[:3]] == [ocode.N, ocode.M, 64] and abs(float(g[3]) - ocode.rate) < 1e-7 and [int(x) for x in g[4:8]] == [6, 576, 2, 4]
    mul, add, inv = nbm.GFInitial(64, os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
    assert [int(x) for x in line["GF"]] == [int(mul[2, 33]), int(add[5, 9]), int(inv[5]), int(mul[7, 9]), 7 ^ 9, int(inv[13])]
    assert [float(x) for x in line["CON"]] == [1.0, 0.0, -1.0, 0.0] and line["BITTOSYM"] == ["1"]
    seed = np.array([173, 173, 173], np.int32)
    rx, _ = orc.nb_channel(ocode, cw, seed, sigma)
    a = line["AWGN"]
    assert [np.float32(a[0]), np.float32(a[1]), np.float32(a[2])] == [rx[0], rx[1], rx[-1]] and int(a[3], 16) == orc.fold_hash(rx)
    assert [int(x) for x in a[4:7]] == seed.tolist()
    st = line["STAT"]
    assert [int(x) for x in st[:8]] == [0, 0, 0, 1, 4, 2, 4, 2 + 3 + 4 + 5]  # returns 1 when >= 2 error frames and >= 3 frames
    assert abs(float(st[8]) - 0.5) < 1e-7 and abs(float(st[9]) - 4 / 4 / 96) < 1e-9 and abs(float(st[10]) - 14 / 4) < 1e-6  # "BER" = symbol errors / frames / N (sic)

