#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference
for the NB part):  python oracle/make_golden.py

NB EMS / TMM / layered TMM  (tests/golden/nb_ref[_tmm|_ltmm]_<snr>dB.npz): outputs of the REFERENCE's own CPU
  decoder (oracle/_ref/nb_ref, compiled from /root/reference/myNBLDPC/src/*.cpp)
  for the first 16 frames of the seed-173 stream at Eb/N0 = 2, 3 and 5 dB:
  channel samples rx, DecodeOutput, iter_number, return flag, fold hashes of
  L_ch / final LLR / final c2v for every frame, and the full L_ch / LLR / c2v
  arrays of a few frames.
binary  (tests/golden/bldpc_*.npz): inputs y and outputs D of OUR restatement
  (oracle/bldpc_oracle.c) whose D-hashes equal the values SURVEY.md 8c recorded
  from a host emulation of the reference kernels; the fixture stores y so a
  libm difference on another box cannot perturb decode parity.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FULL_FRAMES = {2.0: [0], 3.0: [0, 1, 4], 5.0: [0]}


def parse_nb_dump(path):
    b = open(path, "rb").read()
    hdr = np.frombuffer(b[:32], np.int32)
    N, M, q, dv, dc, maxit, nf, method = (int(x) for x in hdr)
    nqam, method = method >> 8, method & 0xFF  # n_QAM when it is not 2: one complex channel sample per code symbol
    nv = q - 1 if method == 0 else q  # the trellis decoders keep element 0 in their vectors
    sigma, rate = np.frombuffer(b[32:40], np.float32)
    off = 40
    cw = np.frombuffer(b[off:off + 4 * N], np.int32).copy(); off += 4 * N
    m = int(np.log2(q))
    recs = []
    for _fr in range(nf):
        r = {}
        nrx = 2 * N if nqam else N * m
        r["rx"] = np.frombuffer(b[off:off + 4 * nrx], np.float32).copy(); off += 4 * nrx
        r["Lch"] = np.frombuffer(b[off:off + 4 * N * (q - 1)], np.float32).reshape(N, q - 1).copy(); off += 4 * N * (q - 1)
        r["out"] = np.frombuffer(b[off:off + 4 * N], np.int32).copy(); off += 4 * N
        r["it"], r["ok"] = (int(x) for x in np.frombuffer(b[off:off + 8], np.int32)); off += 8
        r["LLR"] = np.frombuffer(b[off:off + 4 * N * nv], np.float32).reshape(N, nv).copy(); off += 4 * N * nv
        r["c2v"] = np.frombuffer(b[off:off + 4 * M * dc * nv], np.float32).reshape(M, dc, nv).copy(); off += 4 * M * dc * nv
        recs.append(r)
    assert off == len(b)
    return dict(N=N, M=M, q=q, dv=dv, dc=dc, maxit=maxit, sigma=float(sigma), rate=float(rate), cw=cw, recs=recs)


def make_nb(method=0, tag="nb_ref"):
    """method = the reference's decoder_method: 0 Decoding_EMS, 1 Decoding_TMM, 3 Decoding_layered_TMM."""
    ref = orc.ref_binary()
    if ref is None:
        print("oracle/_ref/nb_ref missing -> NB golden not regenerated")
        return
    for snr in (2.0, 3.0, 5.0):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "d.bin")
            subprocess.check_call([ref, "dump", str(snr), "16", out, str(method)], cwd=os.path.join(ROOT, "data", "nb"),
                                  stdout=subprocess.DEVNULL)
            d = parse_nb_dump(out)
        recs = d["recs"]
        full = FULL_FRAMES[snr]
        np.savez_compressed(
            os.path.join(GOLD, "%s_%gdB.npz" % (tag, snr)),
            snr=np.float32(snr), sigma=np.float32(d["sigma"]), rate=np.float32(d["rate"]), maxit=d["maxit"], cw=d["cw"],
            rx=np.stack([r["rx"] for r in recs]), out=np.stack([r["out"] for r in recs]),
            it=np.array([r["it"] for r in recs], np.int32), ok=np.array([r["ok"] for r in recs], np.int32),
            Lch_hash=np.array([orc.fold_hash(r["Lch"]) for r in recs], np.uint32),
            LLR_hash=np.array([orc.fold_hash(r["LLR"]) for r in recs], np.uint32),
            c2v_hash=np.array([orc.fold_hash(r["c2v"]) for r in recs], np.uint32),
            full_frames=np.array(full, np.int32),
            full_Lch=np.stack([recs[i]["Lch"] for i in full]), full_LLR=np.stack([recs[i]["LLR"] for i in full]),
            full_c2v=np.stack([recs[i]["c2v"] for i in full]))
        print("NB %s %.1f dB: iters" % (tag, snr), [r["it"] for r in recs], "ok", [r["ok"] for r in recs])
    np.savetxt(os.path.join(ROOT, "data", "nb", "codeword_bds_gf64.txt"), d["cw"][None, :], fmt="%d")


def make_nb_exp64():
    """The reference's exponent-format GF(64) file LDPC_N576_K288_GF64_d1_exp.txt, read the way its Get_H reads it (the exponents as
    field elements, zeros included), through its own Decoding_EMS (oracle/_ref/nb_ref_exp64, all-zero codeword): 8 frames per Eb/N0."""
    ref = os.path.join(HERE, "_ref", "nb_ref_exp64")
    if not os.path.exists(ref):
        print("oracle/_ref/nb_ref_exp64 missing -> golden not regenerated")
        return
    for snr in (3.0, 5.0):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "d.bin")
            subprocess.check_call([ref, "dump", str(snr), "8", out, "0"], cwd=os.path.join(ROOT, "data", "nb"), stdout=subprocess.DEVNULL)
            d = parse_nb_dump(out)
        recs = d["recs"]
        assert d["q"] == 64 and d["N"] == 96 and d["M"] == 48
        np.savez_compressed(
            os.path.join(GOLD, "nb_ref_exp64_%gdB.npz" % snr),
            snr=np.float32(snr), sigma=np.float32(d["sigma"]), rate=np.float32(d["rate"]), maxit=d["maxit"], cw=d["cw"],
            rx=np.stack([r["rx"] for r in recs]), out=np.stack([r["out"] for r in recs]),
            it=np.array([r["it"] for r in recs], np.int32), ok=np.array([r["ok"] for r in recs], np.int32),
            Lch_hash=np.array([orc.fold_hash(r["Lch"]) for r in recs], np.uint32),
            LLR=np.stack([r["LLR"] for r in recs]), c2v=np.stack([r["c2v"] for r in recs]))
        print("NB exp64 %.1f dB: iters" % snr, [r["it"] for r in recs], "ok", [r["ok"] for r in recs])


def make_nb_qam64():
    """The n_QAM != 2 branches of the reference (Modulate / AWGNChannel_CPU / Demodulate, LDPC_Encoder.cpp:18-68,
    LDPC_Decoder.cpp:160-169) through the reference itself: oracle/_ref/nb_ref_qam64 = define.h with n_QAM 64 and the Gray 64-QAM
    constellation, BDS code and codeword: 8 frames per Eb/N0 (complex channel samples, L_ch, decode results)."""
    ref = os.path.join(HERE, "_ref", "nb_ref_qam64")
    if not os.path.exists(ref):
        print("oracle/_ref/nb_ref_qam64 missing -> golden not regenerated")
        return
    for snr in (11.0, 14.0):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "d.bin")
            subprocess.check_call([ref, "dump", str(snr), "8", out, "0"], cwd=os.path.join(ROOT, "data", "nb"), stdout=subprocess.DEVNULL)
            d = parse_nb_dump(out)
        recs = d["recs"]
        np.savez_compressed(
            os.path.join(GOLD, "nb_ref_qam64_%gdB.npz" % snr),
            snr=np.float32(snr), sigma=np.float32(d["sigma"]), rate=np.float32(d["rate"]), maxit=d["maxit"], cw=d["cw"],
            rx=np.stack([r["rx"].reshape(-1, 2) for r in recs]), out=np.stack([r["out"] for r in recs]),
            it=np.array([r["it"] for r in recs], np.int32), ok=np.array([r["ok"] for r in recs], np.int32),
            Lch=np.stack([r["Lch"] for r in recs]),
            LLR_hash=np.array([orc.fold_hash(r["LLR"]) for r in recs], np.uint32),
            c2v_hash=np.array([orc.fold_hash(r["c2v"]) for r in recs], np.uint32))
        print("NB 64-QAM %.1f dB: iters" % snr, [r["it"] for r in recs], "ok", [r["ok"] for r in recs])


def make_nb_gf256_qam256():
    """GF(256) code over Gray 256-QAM (one point per symbol) through the reference built with GFQ 256, n_QAM 256: 8 frames per Eb/N0."""
    ref = os.path.join(HERE, "_ref", "nb_ref_gf256_qam256")
    if not os.path.exists(ref):
        print("oracle/_ref/nb_ref_gf256_qam256 missing -> golden not regenerated")
        return
    for snr in (14.0, 18.0):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "d.bin")
            subprocess.check_call([ref, "dump", str(snr), "8", out, "0"], cwd=os.path.join(ROOT, "data", "nb"), stdout=subprocess.DEVNULL)
            d = parse_nb_dump(out)
        recs = d["recs"]
        assert d["q"] == 256
        np.savez_compressed(
            os.path.join(GOLD, "nb_ref_gf256_qam256_%gdB.npz" % snr),
            snr=np.float32(snr), sigma=np.float32(d["sigma"]), rate=np.float32(d["rate"]), maxit=d["maxit"], cw=d["cw"],
            rx=np.stack([r["rx"].reshape(-1, 2) for r in recs]), out=np.stack([r["out"] for r in recs]),
            it=np.array([r["it"] for r in recs], np.int32), ok=np.array([r["ok"] for r in recs], np.int32),
            Lch=np.stack([r["Lch"] for r in recs]),
            LLR_hash=np.array([orc.fold_hash(r["LLR"]) for r in recs], np.uint32),
            c2v_hash=np.array([orc.fold_hash(r["c2v"]) for r in recs], np.uint32))
        print("NB GF(256) 256-QAM %.1f dB: iters" % snr, [r["it"] for r in recs], "ok", [r["ok"] for r in recs])


def make_nb_gf256():
    """The reference's GF(256) code LDPC_N96_K48_GF256_d1_exp.txt (12 symbols, 6 checks) through its own Decoding_EMS
    (oracle/_ref/nb_ref_gf256: define.h's Matrixfile / GFQ edited at build time, all-zero codeword): 12 frames per Eb/N0."""
    ref = os.path.join(HERE, "_ref", "nb_ref_gf256")
    if not os.path.exists(ref):
        print("oracle/_ref/nb_ref_gf256 missing -> GF(256) golden not regenerated")
        return
    for snr in (2.0, 4.0, 6.0):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "d.bin")
            subprocess.check_call([ref, "dump", str(snr), "12", out, "0"], cwd=os.path.join(ROOT, "data", "nb"), stdout=subprocess.DEVNULL)
            d = parse_nb_dump(out)
        recs = d["recs"]
        assert d["q"] == 256 and d["N"] == 12 and d["M"] == 6
        np.savez_compressed(
            os.path.join(GOLD, "nb_ref_gf256_%gdB.npz" % snr),
            snr=np.float32(snr), sigma=np.float32(d["sigma"]), rate=np.float32(d["rate"]), maxit=d["maxit"], cw=d["cw"],
            rx=np.stack([r["rx"] for r in recs]), out=np.stack([r["out"] for r in recs]),
            it=np.array([r["it"] for r in recs], np.int32), ok=np.array([r["ok"] for r in recs], np.int32),
            Lch=np.stack([r["Lch"] for r in recs]), LLR=np.stack([r["LLR"] for r in recs]), c2v=np.stack([r["c2v"] for r in recs]))
        print("NB GF(256) %.1f dB: iters" % snr, [r["it"] for r in recs], "ok", [r["ok"] for r in recs])


def make_nb_heavy_rows():
    """The reference's two codes whose check rows are heavier than 6: Tanner_74_9_Z128_GF16.txt (GF(16), 9472 symbols, 1152 checks,
    dv 3, dc 21) and LDPC_N576_K480_GF256_exp.txt (GF(256), 72 symbols, 12 checks, dv 2, dc 12), through the reference's own
    Decoding_EMS (oracle/_ref/nb_ref_tanner16, nb_ref_gf256_dc12: define.h's Matrixfile / GFQ / maxdc / maxdv edited at build time,
    all-zero codeword).  The Tanner code costs the reference ~2.4 s per iteration and frame: two frames per Eb/N0."""
    for tag, q, N, M, cases in (("tanner16", 16, 9472, 1152, ((5.0, 2), (6.0, 2))), ("gf256_dc12", 256, 72, 12, ((5.0, 4), (7.0, 4)))):
        ref = os.path.join(HERE, "_ref", "nb_ref_" + tag)
        if not os.path.exists(ref):
            print("oracle/_ref/nb_ref_%s missing -> golden not regenerated" % tag)
            continue
        for snr, frames in cases:
            with tempfile.TemporaryDirectory() as td:
                out = os.path.join(td, "d.bin")
                subprocess.check_call([ref, "dump", str(snr), str(frames), out, "0"], cwd=os.path.join(ROOT, "data", "nb"), stdout=subprocess.DEVNULL)
                d = parse_nb_dump(out)
            recs = d["recs"]
            assert d["q"] == q and d["N"] == N and d["M"] == M
            np.savez_compressed(
                os.path.join(GOLD, "nb_ref_%s_%gdB.npz" % (tag, snr)),
                snr=np.float32(snr), sigma=np.float32(d["sigma"]), rate=np.float32(d["rate"]), maxit=d["maxit"], cw=d["cw"].astype(np.uint8),
                rx=np.stack([r["rx"] for r in recs]), out=np.stack([r["out"] for r in recs]).astype(np.uint8),
                it=np.array([r["it"] for r in recs], np.int32), ok=np.array([r["ok"] for r in recs], np.int32),
                Lch_hash=np.array([orc.fold_hash(r["Lch"]) for r in recs], np.uint32),
                LLR_hash=np.array([orc.fold_hash(r["LLR"]) for r in recs], np.uint32),
                c2v_hash=np.array([orc.fold_hash(r["c2v"]) for r in recs], np.uint32))
            print("NB %s %.1f dB: iters" % (tag, snr), [r["it"] for r in recs], "ok", [r["ok"] for r in recs])


# (file, J, L, Z, F, Es/N0 dB, literal table?, hash recorded in SURVEY.md 8c or None)
BIN_CASES = [
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 32, 3.0, False, 0x05A41534),
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 32, 4.0, True, 0x90C5DF9B),
    ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 16, -1.0, False, 0xC00D92C5),
]


def make_binary():
    for name, J, L, Z, F, snr, literal, want in BIN_CASES:
        code = orc.BinaryCode(os.path.join(ROOT, "data", "bldpc", name), J, L, Z, literal=literal)
        seed = np.array([173, 173, 173], np.int32)
        sigma = orc.bldpc_sigma(snr)
        y = orc.bldpc_awgn(seed, sigma, code.N, F)
        r = orc.bldpc_decode(code, y, F, 50, early_exit=1)
        h = orc.fold_hash(r["D"][: code.N * F])
        assert want is None or h == want, "%s %g dB: %08x != %08x" % (name, snr, h, want)
        tag = "%s_%gdB_%s" % (name.split("_BlockH")[0], snr, "lit" if literal else "cor")
        np.savez_compressed(os.path.join(GOLD, "bldpc_%s.npz" % tag), J=J, L=L, Z=Z, F=F, snr=np.float32(snr),
                            sigma=np.float32(sigma), literal=literal, y=y,
                            D_bits=np.packbits(r["D"][: code.N * F].astype(np.uint8)), flags=r["D"][code.N * F:],
                            it=r["it"], hash=np.uint32(h))
        print("binary %s: it=%d hash=%08x" % (tag, r["it"], h))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    orc.build()
    make_binary()
    make_nb()
    make_nb(1, "nb_ref_tmm")
    make_nb(3, "nb_ref_ltmm")
    make_nb_gf256()
    make_nb_exp64()
    make_nb_qam64()
    make_nb_gf256_qam256()
    make_nb_heavy_rows()
