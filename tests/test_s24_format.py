"""The S24 form of a bit matrix (pybmf_amd/csrc/s24.h) against a model of v_smfmac_i32_16x16x128_i8.

The encoder is plain C shared by the device packer; here it is compiled with gcc and checked, without a GPU, against (i) the operand
layout of the instruction as measured by scripts/probes/smfmac_probe.hip (gpurun_out of round 2: A slot pair (2j, 2j+1) of lane group a
selects among bytes 4j..4j+3 of B lane groups 2 (a & 1) / 2 (a & 1) + 1, bytes 16 (a >> 1) ..), (ii) the digit-plane order of the dense
kernel (bmf_panel_pos_i8), which the sparse kernel reads with the same two 16-byte fragments per lane.  The GPU tests
(tests/test_s24_gpu.py) then check the device packer against this encoder bit for bit and the kernel against the dense one."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = tmp_path_factory.mktemp("s24") / "s24_shim.so"
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tests", "csrc", "s24_shim.c"), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    lib.s24_code.restype = C.c_uint
    lib.s24_encode_pair.restype = C.c_uint
    return lib


def panel_pos_i8(cl):
    """bmf_panel_pos_i8_dev (pybmf_amd/csrc/common.h)"""
    g, t, bit = cl >> 7, (cl >> 5) & 3, cl & 31
    b, s = bit >> 3, bit & 7
    return 128 * t + ((s >> 2) * 4 + g) * 16 + 4 * (s & 3) + b


def encode_block(shim, words):
    """words[16] (word 4 g + t of a 512-block) -> idx[a][t], val[a][tp], kept[16], extra"""
    idx = np.zeros((4, 4), np.uint32)
    val = np.zeros((4, 2), np.uint32)
    kept = np.zeros(16, np.uint32)
    extra = 0
    for h in range(2):
        w = np.array([[words[4 * (2 * h + gi) + t] for t in range(4)] for gi in range(2)], np.uint32)
        i2 = np.zeros((2, 4), np.uint32)
        v2 = np.zeros((2, 2), np.uint32)
        k2 = np.zeros((2, 4), np.uint32)
        extra += shim.s24_encode_pair(w.ctypes.data_as(C.c_void_p), i2.ctypes.data_as(C.c_void_p), v2.ctypes.data_as(C.c_void_p),
                                      k2.ctypes.data_as(C.c_void_p))
        for ai in range(2):
            idx[h + 2 * ai] = i2[ai]
            val[h + 2 * ai] = v2[ai]
        for gi in range(2):
            for t in range(4):
                kept[4 * (2 * h + gi) + t] = k2[gi][t]
    return idx, val, kept, extra


def smfmac_model(idx, val, plane):
    """One row x one column over a 512-block: sum over the four stages of what the instruction adds, operands as the kernel feeds them."""
    total = 0
    for t in range(4):
        for a in range(4):
            vw = int(val[a][t >> 1])
            iw = int(idx[a][t])
            u = t & 1
            for sg in range(16):
                v = (vw >> (4 * u + (sg >> 2) + 8 * (sg & 3))) & 1
                p = (iw >> (2 * sg)) & 3
                bg = 2 * (a & 1) + (sg >> 3)
                y = 16 * (a >> 1) + 4 * ((sg >> 1) & 3) + p
                # B lane group bg: bytes 0..15 = chunk (ks = 0, g = bg) of stage t, bytes 16..31 = chunk (ks = 1, g = bg)
                byte = plane[128 * t + ((y >> 4) * 4 + bg) * 16 + (y & 15)]
                total += v * int(byte)
    return total


def test_code_table(shim):
    for m in range(16):
        c = shim.s24_code(m)
        p0, p1, v0, v1, extra = c & 3, (c >> 2) & 3, (c >> 4) & 1, (c >> 5) & 1, c >> 6
        assert p0 < p1
        ones = [b for b in range(4) if (m >> b) & 1]
        kept = ([p0] if v0 else []) + ([p1] if v1 else [])
        assert kept == ones[:2]
        assert extra == max(0, len(ones) - 2)


@pytest.mark.parametrize("density", [0.02, 0.08, 0.3, 0.7, 1.0])
def test_encoded_block_reproduces_the_kept_ones(shim, density):
    rs = np.random.RandomState(int(density * 100))
    pos = np.array([panel_pos_i8(cl) for cl in range(512)])
    assert sorted(pos.tolist()) == list(range(512))
    for _ in range(12):
        bits = (rs.rand(512) < density).astype(np.uint8)          # bits[cl], cl = 128 g + 32 t + bit
        words = np.zeros(16, np.uint32)
        for cl in np.nonzero(bits)[0]:
            g, t, bit = cl >> 7, (cl >> 5) & 3, cl & 31
            words[4 * g + t] |= np.uint32(1) << np.uint32(bit)
        idx, val, kept, extra = encode_block(shim, words)
        kept_bits = np.zeros(512, np.uint8)
        for cl in range(512):
            g, t, bit = cl >> 7, (cl >> 5) & 3, cl & 31
            kept_bits[cl] = (int(kept[4 * g + t]) >> bit) & 1
        assert np.all(kept_bits <= bits) and int(bits.sum()) - int(kept_bits.sum()) == extra
        # every group of four {s, s+8, s+16, s+24} keeps its first two ones
        for wi in range(16):
            for s in range(8):
                o = [b for b in range(4) if (int(words[wi]) >> (s + 8 * b)) & 1]
                k = [b for b in range(4) if (int(kept[wi]) >> (s + 8 * b)) & 1]
                assert k == o[:2]
        f = rs.randint(-128, 128, size=512)                        # digits of one factor column, by reduction index
        plane = np.zeros(512, np.int64)
        plane[pos] = f
        assert smfmac_model(idx, val, plane) == int((kept_bits.astype(np.int64) * f).sum())
