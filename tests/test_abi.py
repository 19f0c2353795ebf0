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


def test_cpp_adapter_loads_the_replay_drivers_map_files(built, maps, tmp_path):
    """mylsd::loadMapParam / loadMapValue (include/myLSD.h): the text formats LSD/main_on_windows.cpp:27-46 reads -- "cols rows resolution
    originX originY" and rows x cols decimal cell values -- give the adapter's caller the same bytes as the fixtures (which
    tests/golden/make_fixtures.py read from the reference's data/*.txt the same way); values above 255 keep their low byte, as the
    driver's fscanf("%d") into a uint8_t does."""
    img = maps["map1"][:40, :56].copy()
    txt = tmp_path / "mapValue.txt"
    vals = img.astype(np.int64)
    vals[3, 5] += 256                                             # (a cell value with more than 8 bits)
    txt.write_text("\n".join(" ".join(str(int(v)) for v in row) for row in vals) + "\n")
    (tmp_path / "mapParam.txt").write_text("56 40 0.025 -4.43187 -5.49357\n")
    src = tmp_path / "t.cpp"
    src.write_text(
        '#include "myLSD.h"\n'
        '#include <cstdio>\n'
        'int main(int argc, char** argv) {\n'
        '  structMapParam mp; mylsd::Mat m;\n'
        '  if (!mylsd::loadMapParam(argv[1], &mp) || !mylsd::loadMapValue(argv[2], mp.oriMapCol, mp.oriMapRow, &m)) return 2;\n'
        '  if (mylsd::loadMapValue(argv[1], mp.oriMapCol, mp.oriMapRow, &m)) return 3;      /* a short file is refused */\n'
        '  std::printf("%d %d %.5f %.5f %.5f\\n", mp.oriMapCol, mp.oriMapRow, mp.mapResol, mp.mapOriX, mp.mapOriY);\n'
        '  for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.cols; c++) std::printf("%d\\n", (int)m.ptr<unsigned char>(r)[c]);\n'
        '  return 0;\n'
        '}\n')
    exe = tmp_path / "t"
    pkg = os.path.join(ROOT, "linesegmentdetector-slam_amd")
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", pkg, "-llsdhip", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    p = subprocess.run([str(exe), str(tmp_path / "mapParam.txt"), str(txt)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.split("\n")
    assert lines[0] == "56 40 0.02500 -4.43187 -5.49357"
    got = np.array([int(x) for x in lines[1:] if x], np.uint8).reshape(40, 56)
    assert np.array_equal(got, img)


# A header with the include guard and the colliding names of the reference's LSD/baseFunc.h (:20-88): what a caller that
# has the reference tree sees under that file name.  (Test double written here: the real one pulls in Eigen, absent in
# this image; only the guard and the names matter for the collision.)
_BASEFUNC_DOUBLE = '''
#ifndef _BASEFUNC_
#define _BASEFUNC_
#define TEST_SAW_REFERENCE_BASEFUNC 1
typedef struct _structMapParam { int oriMapCol; int oriMapRow; double mapResol; double mapOriX; double mapOriY; } structMapParam;
typedef struct _structLinesInfo { double k; double b; double dx; double dy; double x1; double y1; double x2; double y2; double len; int orient; } structLinesInfo;
typedef struct _structPosition { double x; double y; double ang; } structPosition;
double sind(double x); double cosd(double x); double atand(double x);
const double z_occ_max_dis = 1;
const double lsd_sca = 0.3; const double lsd_sig = 0.6; const double lsd_angThre = 22.5; const double lsd_denThre = 0.7;
const int pseBin = 1024;
const int maxEstiDist = 60;
#endif
'''

_CALLER = '''
%s
int main() {
  structMapParam mp; mp.oriMapCol = 608;
  structPosition pose; pose.x = 0;
  mylsd::structLSD (*f)(mylsd::Mat, int, int, double, double, double, double, int) = &mylsd::myLineSegmentDetector;
  static_assert(sizeof(structLinesInfo) == 80, "layout");
  return (f != nullptr && mp.oriMapCol == 608 && pose.x == 0 && pseBin == 1024 && lsd_sca == 0.3 && z_occ_max_dis == 1) ? 0 : 1;
}
'''


@pytest.mark.parametrize("case", ["on_include_path", "quoted_later_only", "basefunc_first", "no_reference_tree"])
def test_cpp_adapter_in_the_reference_include_order(case, tmp_path):
    """LSD/main_on_windows.cpp:5-8 includes <myLSD.h> first and <baseFunc.h> afterwards; LSD/myLSD.h:37 includes
    <baseFunc.h> itself.  The adapter must compile in that order whether baseFunc.h is on the include path (it is then
    included by the adapter, like the reference header does), only reachable later through a quoted include (the adapter
    has then defined the same guard), included before the adapter, or absent."""
    inc = os.path.join(ROOT, "include")
    ref = tmp_path / "reftree"
    ref.mkdir()
    src_dir = tmp_path / "src"
    src_dir.mkdir()
    flags = ["-I", inc]
    if case == "on_include_path":
        (ref / "baseFunc.h").write_text(_BASEFUNC_DOUBLE)
        flags += ["-I", str(ref)]
        head = "#include <myLSD.h>\n#include <baseFunc.h>\n#ifndef TEST_SAW_REFERENCE_BASEFUNC\n#error the adapter must take the reference baseFunc.h when it is on the include path\n#endif\n#ifdef LSD_ADAPTER_OWN_BASEFUNC\n#error fallback active although baseFunc.h is reachable\n#endif"
    elif case == "quoted_later_only":
        (src_dir / "baseFunc.h").write_text(_BASEFUNC_DOUBLE)
        head = '#include <myLSD.h>\n#include "baseFunc.h"\n#ifndef LSD_ADAPTER_OWN_BASEFUNC\n#error expected the fallback declarations\n#endif'
    elif case == "basefunc_first":
        (ref / "baseFunc.h").write_text(_BASEFUNC_DOUBLE)
        flags += ["-I", str(ref)]
        head = "#include <baseFunc.h>\n#include <myLSD.h>\n#include <baseFunc.h>"
    else:
        head = "#include <myLSD.h>\n#ifndef LSD_ADAPTER_OWN_BASEFUNC\n#error expected the fallback declarations\n#endif"
    src = src_dir / "caller.cpp"
    src.write_text(_CALLER % head)
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror"] + flags + [str(src)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_cpp_adapter_wins_over_a_second_myLSD_h_on_the_path(tmp_path):
    """The reference compiles with -I LSD and includes <myLSD.h> with angle brackets (LSD/main_on_windows.cpp:5), and the
    adapter has the same file name: -I<repo>/include must come BEFORE -I LSD.  With both
    directories on the path the adapter is the one that is found first, takes the reference's baseFunc.h from the second
    directory, and the reference's own myLSD.h (a stand-in here: it would need OpenCV) is never opened; in the wrong order
    the build fails loudly."""
    inc = os.path.join(ROOT, "include")
    ref = tmp_path / "LSD"
    ref.mkdir()
    (ref / "baseFunc.h").write_text(_BASEFUNC_DOUBLE)
    (ref / "myLSD.h").write_text("#error the reference's LSD/myLSD.h was picked up: put -I<repo>/include before -I LSD\n")
    src = tmp_path / "caller.cpp"
    src.write_text(_CALLER % "#include <myLSD.h>\n#include <baseFunc.h>\n#ifndef TEST_SAW_REFERENCE_BASEFUNC\n#error baseFunc.h of the reference tree expected\n#endif")
    ok = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", inc, "-I", str(ref), str(src)], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    bad = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", str(ref), "-I", inc, str(src)], capture_output=True, text=True)
    assert bad.returncode != 0 and "put -I<repo>/include before -I LSD" in bad.stderr


def test_cpp_adapter_opencv_branch_compiles(tmp_path):
    """The LSD_WITH_OPENCV branch (cv::Mat in the signatures, as the reference's callers pass it) through a compiler:
    against a test double of <opencv2/core.hpp> with cv::Mat's surface that the adapter uses (zeros -> MatExpr, ptr<T>(row),
    rows, cols, step as cv::MatStep).  OpenCV itself is not in this image."""
    cvdir = tmp_path / "cv" / "opencv2"
    cvdir.mkdir(parents=True)
    (cvdir / "core.hpp").write_text('''
#pragma once
#include <cstddef>
#include <cstdlib>
#include <memory>
#define CV_8UC1 0
#define CV_64FC1 6
namespace cv {
struct MatStep { size_t v; MatStep(size_t s = 0) : v(s) {} operator size_t() const { return v; } };
class Mat; struct MatExpr { int r, c, t; operator Mat() const; };
class Mat {
public:
    int rows = 0, cols = 0; unsigned char* data = nullptr; MatStep step;
    Mat() {}
    static MatExpr zeros(int r, int c, int t) { return MatExpr{r, c, t}; }
    template <class T> T* ptr(int row = 0) { return reinterpret_cast<T*>(data + (size_t)row * (size_t)step); }
    template <class T> const T* ptr(int row = 0) const { return reinterpret_cast<const T*>(data + (size_t)row * (size_t)step); }
    void release() { buf.reset(); data = nullptr; }
    std::shared_ptr<unsigned char> buf;
};
inline MatExpr::operator Mat() const {
    Mat m; m.rows = r; m.cols = c; const size_t es = t == CV_64FC1 ? 8 : 1; m.step = MatStep(es * (size_t)c);
    m.buf.reset(static_cast<unsigned char*>(std::calloc((size_t)r * es * (size_t)c + 16, 1)), std::free); m.data = m.buf.get(); return m;
}
}
''')
    src = tmp_path / "caller.cpp"
    src.write_text('''
#include <opencv2/core.hpp>
#include <myLSD.h>
using namespace cv;
#ifndef LSD_WITH_OPENCV
#error the adapter did not pick up OpenCV
#endif
int main() {
  Mat mapValue = Mat::zeros(480, 608, CV_8UC1);
  static_assert(sizeof(mylsd::Mat) == sizeof(cv::Mat), "mylsd::Mat is cv::Mat");
  try {
    Mat mapCache = mylsd::createMapCache(mapValue, 0.05);                      /* LSD/main_on_windows.cpp:67 */
    mylsd::structLSD LSD = mylsd::myLineSegmentDetector(mapValue, 608, 480, lsd_sca, lsd_sig, lsd_angThre, lsd_denThre, pseBin);   /* :70 */
    return LSD.len_linesInfo < 0;
  } catch (const mylsd::lsd_error& e) { return e.status == LSD_ERR_NO_DEVICE ? 0 : 1; }
}
''')
    pkg = os.path.join(ROOT, "linesegmentdetector-slam_amd")
    exe = tmp_path / "t"
    p = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", str(tmp_path / "cv"), "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", pkg, "-llsdhip", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([str(exe)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr      # no GPU here: lsd_error(NO_DEVICE) through the cv::Mat signatures


def test_region_kernel_keeps_its_registers_and_stays_out_of_scratch(tmp_path):
    """The region stage's out-of-line stages get their context from LDS: a struct passed by value travels through scratch
    memory at every call (it did: 1840 bytes per lane, 7 % of the stage's time).  The compiler's own resource summary of the
    8-wave kernel must stay at two waves per SIMD (<= 256 VGPRs) with a small scratch frame: return-address saves, the
    callee-saved registers eval_seed() -- a non-leaf function that uses the whole register file -- has to put away once per full
    evaluation, and a handful of spilled scalars of the seed loop."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "linesegmentdetector-slam_amd", "csrc", "k_region.hip")
    out = str(tmp_path / "k_region_w8.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-DLSD_REGION_NW=8",
                    "-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
    text = open(out).read()
    kern = text[text.index("; Kernel info:"):]
    vgprs = int(re.search(r"; NumVgprs: (\d+)", kern).group(1))
    scratch = int(re.search(r"; ScratchSize: (\d+)", kern).group(1))
    occ = int(re.search(r"; Occupancy: (\d+)", kern).group(1))
    assert vgprs <= 256 and occ >= 2, (vgprs, occ)
    # (416-448 B since round 6: the 8-wave kernel's body sits in the loop of the persistent workgroups -- k_region.hip: k_region -- and keeps
    #  a few more of the launch's arguments alive around it: 29 scratch stores / 40 loads in the kernel body, none of them per
    #  evaluation; 272 B before.  The image's code as an out-of-line function instead: 736 B -- a function that uses the whole
    #  register file saves every callee-saved register once per image.)
    assert scratch <= 470, scratch                             # (448 B measured)
    # the 4-wave build (throughput mode) is compiled for THREE workgroups per CU: 3 waves per SIMD (<= 168 registers), LDS <= 160 KB / 3
    out4 = str(tmp_path / "k_region_w4.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-DLSD_REGION_NW=4",
                    "-DLSD_REGION_WAVES_PER_SIMD=3", "-S", "--cuda-device-only", "-o", out4, src], check=True, capture_output=True)
    text = open(out4).read()
    kern = text[text.index("; Kernel info:"):]
    assert int(re.search(r"; NumVgprs: (\d+)", kern).group(1)) <= 168 and int(re.search(r"; Occupancy: (\d+)", kern).group(1)) >= 3
    assert int(re.search(r"; LDSByteSize: (\d+)", kern).group(1)) * 3 <= 160 * 1024
    # (592 B: the stages are capped at 168 registers of which the calling convention leaves 104 free of a save; grow() needs ~146 at once
    #  and so saves 43 callee-saved registers per call whether or not it calls anything -- DESIGN.md section 5, "K4's HBM traffic")
    assert int(re.search(r"; ScratchSize: (\d+)", kern).group(1)) <= 620
