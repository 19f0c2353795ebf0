"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/lsd_hip.h declares, keeps the reference's struct layouts, and refuses to run without a GPU
(no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    return g


def _header_functions():
    src = open(os.path.join(ROOT, "include", "lsd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(lsd_[a-z_0-9]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_exported(built, lsdmod):
    lib = lsdmod.load_library()
    names = _header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "liblsdhip.so does not export %s" % n
    assert sorted(lsdmod.EXPORTED_SYMBOLS) == names


def test_exports_are_c_linkage(built):
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "linesegmentdetector-slam_amd", "liblsdhip.so")],
                         capture_output=True, text=True, check=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if l.strip()}
    for n in _header_functions():
        assert n in syms


def test_struct_layouts_match_reference(built, lsdmod):
    # structLinesInfo (LSD/baseFunc.h:33-44): 9 doubles + int, 80 bytes with tail padding
    assert C.sizeof(lsdmod.lsd_line) == 80
    offs = {f: getattr(lsdmod.lsd_line, f).offset for f, _ in lsdmod.lsd_line._fields_}
    assert [offs[k] for k in ("k", "b", "dx", "dy", "x1", "y1", "x2", "y2", "len", "orient")] == list(range(0, 80, 8))
    assert lsdmod.LINE_DTYPE.itemsize == 80
    assert C.sizeof(lsdmod.lsd_params) == 40


def test_defaults_are_basefunc_constants(built, lsdmod):
    p = lsdmod.lsd_params()
    lsdmod.load_library().lsd_default_params(C.byref(p))
    assert (p.sca, p.sig, p.angThre, p.denThre, p.pseBin) == (0.3, 0.6, 22.5, 0.7, 1024)   # LSD/baseFunc.h:64-68
    assert (lsdmod.lsd_sca, lsdmod.lsd_sig, lsdmod.lsd_angThre, lsdmod.lsd_denThre, lsdmod.pseBin) == (0.3, 0.6, 22.5, 0.7, 1024)


def test_scaled_size(built, lsdmod):
    # myLSD.cpp:132-133
    assert lsdmod.scaled_size(608, 480) == (182, 144)
    assert lsdmod.scaled_size(1621, 625) == (486, 187)
    assert lsdmod.scaled_size(2048, 2048) == (614, 614)


def test_strerror_and_version(built, lsdmod):
    lib = lsdmod.load_library()
    assert lib.lsd_abi_version() == 1
    assert lib.lsd_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.lsd_strerror(lsdmod.LSD_ERR_NO_DEVICE)


def test_no_gpu_means_loud_failure(built, lsdmod):
    """On a box without a GPU lsd_create must fail (status NO_DEVICE): the product never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lsdmod.LsdError) as e:
        lsdmod.Context(0)
    assert e.value.status == lsdmod.LSD_ERR_NO_DEVICE
    with pytest.raises(lsdmod.LsdError):
        lsdmod.runLSD(np.zeros((64, 64), np.uint8))


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or open it."""
    pkg = os.path.join(ROOT, "linesegmentdetector-slam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for l in txt.splitlines():
                    code = l.split("//")[0].split("#")[0] if not f.endswith(".py") else l.split("#")[0]
                    assert "lsd_oracle" not in code and "import oracle" not in code and "from oracle" not in code, (f, l)
    out = subprocess.run(["ldd", os.path.join(pkg, "liblsdhip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_cpp_adapter_compiles(built, tmp_path):
    """include/myLSD.h (the adapter with the reference's own names) compiles against its own image type and links."""
    src = tmp_path / "t.cpp"
    src.write_text(
        '#include "myLSD.h"\n'
        '#include <cstdio>\n'
        'int main() {\n'
        '  lsd::Image<unsigned char> m = lsd::Image<unsigned char>::zeros(64, 80);\n'
        '  static_assert(sizeof(structLinesInfo) == 80, "layout");\n'
        '  try { mylsd::structLSD r = mylsd::runLSD(m); std::printf("lines %d\\n", r.len_linesInfo); }\n'
        '  catch (const mylsd::lsd_error& e) { std::printf("status %d\\n", e.status); return e.status == LSD_ERR_NO_DEVICE ? 0 : 1; }\n'
        '  return 0;\n'
        '}\n')
    exe = tmp_path / "t"
    pkg = os.path.join(ROOT, "linesegmentdetector-slam_amd")
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", pkg, "-llsdhip", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
