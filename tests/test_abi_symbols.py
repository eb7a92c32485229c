"""CPU checks of the C-ABI boundary: the library builds in-tree, loads, and exports every
entry point include/radmmm_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "radmmm_hip.h")
LIB = os.path.join(ROOT, "rad_mmm_amd", "libradmmm_hip.so")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(radmmm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("radmmm_rowgemm_f32", "radmmm_wgrad_f32", "radmmm_weightnorm_fwd", "radmmm_weightnorm_bwd",
                 "radmmm_affine_coupling_fwd", "radmmm_affine_coupling_bwd", "radmmm_masked_reduce",
                 "radmmm_pq_spline_fwd", "radmmm_pq_spline_bwd", "radmmm_attn_fwd", "radmmm_attn_bwd",
                 "radmmm_mas_width1", "radmmm_stft_mel", "radmmm_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    lib.radmmm_abi_version.restype = ctypes.c_int
    assert lib.radmmm_abi_version() == 1
    lib.radmmm_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.radmmm_last_error(), bytes)


def test_argument_validation_without_gpu():
    """Descriptor validation happens before any HIP call, so it is testable on CPU."""
    lib = ctypes.CDLL(LIB)
    lib.radmmm_last_error.restype = ctypes.c_char_p
    assert lib.radmmm_rowgemm_f32(None, None) == -1
    assert b"null descriptor" in lib.radmmm_last_error()
    assert lib.radmmm_wgrad_f32(None, None) == -1


def test_python_binding_matches_struct_layout():
    import rad_mmm_amd._lib as L
    # field order/size of the ctypes mirrors == the C structs (checked via a tiny C probe is not
    # possible without hipcc at test time; check the sizes the compiler is known to produce)
    assert ctypes.sizeof(L.RowGemmDesc) % 8 == 0 and ctypes.sizeof(L.WgradDesc) % 8 == 0
    assert L.RowGemmDesc.a_item_stride.offset == 16 and L.RowGemmDesc.B.offset == 24


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "rad_mmm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn
    bench = open(os.path.join(ROOT, "bench.py")).read() if os.path.exists(os.path.join(ROOT, "bench.py")) else ""
    assert "rad_mmm_amd" in bench or bench == ""
