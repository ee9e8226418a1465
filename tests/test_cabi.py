"""CPU: the C-ABI shared library loads and exports every symbol include/bwagpu.h declares; without a GPU it fails
loudly (BWAGPU_ENODEV) instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import pytest

from bwa_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return C.CDLL(build.build(verbose=False))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "bwagpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bwagpu_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported(lib):
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bwagpu.h but not exported by libbwagpu.so"
    assert set(api.EXPORTS) <= set(names)


def test_struct_sizes_in_header_match_python_mirrors():
    from bwa_amd.structs import MemOpt, ALNREG_DTYPE
    assert C.sizeof(MemOpt) == 168 and ALNREG_DTYPE.itemsize == 88
    # the ctypes mirrors against what the library itself was compiled with
    from bwa_amd.index import Built
    L = C.CDLL(api.DEFAULT_LIB)
    sz = (C.c_int32 * 8)()
    L.bwagpu_abi_sizes(sz)
    assert list(sz) == [C.sizeof(MemOpt), ALNREG_DTYPE.itemsize, C.sizeof(api.Stats), C.sizeof(api.IndexDesc), api.CIGAR_DTYPE.itemsize, api.MATESW_DTYPE.itemsize, 48, C.sizeof(Built)], list(sz)


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    lib.bwagpu_create_from_files.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
    rc = lib.bwagpu_create_from_files(C.byref(h), os.path.join(ROOT, "tests", "golden", "g200k").encode(), 0)
    assert rc == -1 and not h.value   # BWAGPU_ENODEV
    lib.bwagpu_strerror.restype = C.c_char_p
    assert b"HIP device" in lib.bwagpu_strerror(rc)
