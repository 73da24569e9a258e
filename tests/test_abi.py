"""The C-ABI library: every symbol declared in include/bmf_hip.h is exported and bound, the ctypes mirrors of the two
structs have the C layout, and bad arguments are refused with an error code and a message (no compute, no GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bmf_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmf_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from pybmf_amd import _lib as L
    names = declared_functions()
    assert len(names) >= 20
    raw = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} is declared in bmf_hip.h but missing from libbmf_hip.so"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in pybmf_amd/_lib.py"
    assert sorted(L.SIGNATURES) == names


def test_struct_layout_matches_c(tmp_path):
    from pybmf_amd import _lib as L
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(bmf_epilogue_args), offsetof(bmf_epilogue_args, stop), sizeof(bmf_penalty_state),'
                   'offsetof(bmf_penalty_state, sum_x), offsetof(bmf_penalty_state, thr_u), offsetof(bmf_penalty_state, log),'
                   'sizeof(bmf_palm_args), offsetof(bmf_palm_args, beta), offsetof(bmf_palm_args, partials),'
                   'sizeof(bmf_palm_state), offsetof(bmf_palm_state, Xtiled), offsetof(bmf_palm_state, beta));return 0;}\n'
                   % HEADER)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(L.EpilogueArgs), L.EpilogueArgs.stop.offset, C.sizeof(L.PenaltyState), L.PenaltyState.sum_x.offset,
            L.PenaltyState.thr_u.offset, L.PenaltyState.log.offset, C.sizeof(L.PalmArgs), L.PalmArgs.beta.offset, L.PalmArgs.partials.offset,
            C.sizeof(L.PalmState), L.PalmState.Xtiled.offset, L.PalmState.beta.offset]
    assert got == want


def test_bad_arguments_are_refused_without_a_gpu():
    from pybmf_amd import _lib as L
    lib = L.lib
    assert lib.bmf_version() == 501 and lib.bmf_struct_bytes(0) > 0 and lib.bmf_struct_bytes(99) == -1
    assert lib.bmf_xf_bits(None, 512, 4, 4, None, 128, 3, 64, None, 512 * 64, 1, None) == -1
    assert b"null pointer" in lib.bmf_last_error()
    assert lib.bmf_xf_bits_slots(500, 4, 3, 64) == -1            # rows_pad not a multiple of 512
    assert lib.bmf_xf_bits_slots(512, 4, 3, 48) == -1            # kp must be 32 or 64
    assert lib.bmf_xf_bits_slots(100352, 640, 3, 64) >= 1
    assert lib.bmf_gram_partial(None, 512, 64, 64, None, 4, None) == -1
    assert lib.bmf_mu_epilogue(None, None) == -1
    st = L.PenaltyState()
    assert lib.bmf_penalty_prepare(C.byref(st), None) == -1 and b"struct_bytes" in lib.bmf_last_error()
    with pytest.raises(L.BmfError, match="struct_bytes"):
        L.check(lib.bmf_penalty_update(C.byref(st), 1.0, None), "bmf_penalty_update")
    assert sorted(lib.bmf_panel_pos(i) for i in range(128)) == list(range(128))
    assert lib.bmf_panel_pos(128) == -1


def test_no_cpu_fallback():
    """Without a GPU the product refuses to run instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    from pybmf_amd.engine import BitMatrix
    from pybmf_amd.models import BinaryMFPenalty
    with pytest.raises(RuntimeError, match="no CPU path"):
        BitMatrix(np.zeros((8, 8), np.uint8), "cuda:0")
    with pytest.raises(RuntimeError, match="no CPU path"):
        BinaryMFPenalty(k=2, init_method="normal", seed=0).fit(np.eye(8, dtype=np.uint8), task="reconstruction",
                                                              show_logs=False, show_result=False, save_model=False)


def test_argument_checks_under_host_asan():
    """the C-ABI's argument-checking layer built with -fsanitize=address on the HOST side (CPU only) and driven
    through every entry point with arguments it must refuse before touching a GPU (tests/asan/abi_asan_driver.c)."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        pytest.skip("needs the ROCm clang for the sanitizer runtime")
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "pybmf_amd", "csrc"), "-j", "8", "asan"], capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-1500:])
    assert "0 entry point(s) not refused" in res.stdout and "AddressSanitizer" not in res.stderr
