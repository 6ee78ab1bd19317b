"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly what include/*.h declares,
and the ctypes table of the host side matches the header (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "conan_fgw_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|long long)\s+(conan_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",")])
        decls[m.group(2)] = n
    return decls


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from conan_fgw_amd import _lib
    return _lib


def test_header_declares_entry_points():
    d = _declared()
    assert len(d) >= 20
    for must in ("conan_radius_graph_csr", "conan_cfconv_fwd", "conan_fgw_barycenter_fwd", "conan_linear_fwd"):
        assert must in d


def test_library_exports_every_declared_symbol(built):
    L = ctypes.CDLL(built.library_path())
    for name in _declared():
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    hdr = open(HEADER).read()
    want = int(re.search(r"#define\s+CONAN_FGW_ABI_VERSION\s+(\d+)", hdr).group(1))
    from conan_fgw_amd import _lib
    assert L.conan_abi_version() == want == _lib.ABI_VERSION


def test_ctypes_table_matches_header(built):
    d = _declared()
    assert set(built.SIGNATURES) == set(d)
    for name, (_res, args) in built.SIGNATURES.items():
        assert len(args) == d[name], name


def test_params_struct_layout(built):
    assert ctypes.sizeof(built.FgwParams) == 48          # 12 x 4-byte fields, matches conan_fgw_params


def test_workspace_queries_need_no_gpu(built):
    L = built.lib()
    assert L.conan_fgw_workspace_bytes(256, 5, 33, 64) > 0
    assert L.conan_linear_wgrad_ws(1000, 128, 128) >= 128 * 128


def test_bad_arguments_are_rejected_without_launching(built):
    L = built.lib()
    assert L.conan_cfconv_fwd(None, None, None, None, None, 10, 128, None, None, None) == -1
    assert L.conan_linear_fwd(None, None, None, None, 4, 4, 4, 0, 0, None, None, None) == -1
