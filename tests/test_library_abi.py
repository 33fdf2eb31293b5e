"""CPU: the C-ABI library loads and exports every symbol include/ba_hip.h declares
(no compute calls -- there is no GPU here), and the ctypes structs match the header."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from bundle_adjustment_amd import hip_backend
    return hip_backend.load_library()


def test_every_declared_symbol_is_exported(lib):
    from bundle_adjustment_amd import hip_backend
    hdr = open(os.path.join(ROOT, "include", "ba_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+char\*|int)\s+(ba_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(hip_backend.SYMBOLS), declared ^ set(hip_backend.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    import ctypes as C
    from bundle_adjustment_amd import hip_backend as hb
    assert C.sizeof(hb.BAOptions) == 2 * 4 + 6 * 8 + 8 * 4 + 8 + 2 * 4
    assert C.sizeof(hb.BASummary) == 4 * 4 + 9 * 8
    assert C.sizeof(hb.BAProfile) == 2 * (16 * 4 + 16 * 8)
    hdr = open(os.path.join(ROOT, "include", "ba_hip.h")).read()
    assert C.sizeof(hb.BAIterRecord) == 4 * 4 + 7 * 8
    for struct, cls in (("ba_options", hb.BAOptions), ("ba_summary", hb.BASummary), ("ba_iter_record", hb.BAIterRecord)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, flags=re.S).group(1)
        names = re.findall(r"(?:int32_t|double)\s+([a-z_0-9]+)\s*;", body)
        assert names == [n.rstrip("_") for n, _ in cls._fields_]          # (lambda_ <-> lambda: a Python keyword)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bundle_adjustment_amd import hip_backend as hb
    monkeypatch.setattr(hb, "_lib", None)
    monkeypatch.setattr(hb, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(hb.BAHipError):
        hb.load_library()


def test_error_string_and_kernel_names(lib):
    assert lib.ba_kernel_name(5) == b"schur_pt" and lib.ba_kernel_name(6) == b"schur_cam"
    assert lib.ba_destroy(None) == 0
    assert lib.ba_synchronize(None) == -1 and b"null handle" in lib.ba_last_error()
