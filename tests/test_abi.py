"""The C-ABI library: loads on a CPU-only box, exports exactly what include/rsm.h declares, and refuses to
run without a GPU (no CPU fallback anywhere in the product path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rsm.h")


def declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsm_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_what_the_binding_lists():
    from reconstruction_amd import _lib
    assert declared() == sorted(_lib.EXPORTS)


def test_library_loads_and_exports_every_declared_symbol():
    from reconstruction_amd import _lib
    lib = _lib.load()
    for name in declared():
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (rsm_[a-z0-9_]+)", out))
    assert exported == set(declared()), exported ^ set(declared())
    assert lib.rsm_version().startswith(b"rsm-mi355")
    assert lib.rsm_profile_stage_count() >= 10


def test_struct_layouts_match_the_header():
    from reconstruction_amd import _lib
    # rsm_boundary = 6 ints; rsm_pair_in: 4 ptr + 4 int + (pad) double + 2 int + 28 double + int
    assert C.sizeof(_lib.Boundary) == 24
    assert C.sizeof(_lib.PairIn) == 4 * 8 + 4 * 4 + 8 + 2 * 4 + 28 * 8 + 8
    assert C.sizeof(_lib.PairOut) == 2 * 8 + 2 * 24 + 8 + 8 + 8 + 8 + 8 + 8


def test_struct_layouts_as_the_c_compiler_sees_them(tmp_path):
    """sizeof / offsetof of the header's structs from gcc itself against the ctypes mirror (rsm_pair_out grew a trailing
    `points16` in round 5: both sides must agree on every member's offset)."""
    from reconstruction_amd import _lib
    src = tmp_path / "layout.c"
    fields = {"rsm_pair_in": ["image", "mask", "width", "height", "pyr_levels", "radius", "ws", "offset", "origin_width", "Q", "R_final", "T_final", "verbose"],
              "rsm_pair_out": ["disparity", "margin", "n_points", "max_points", "xyz", "bgr", "v_top", "points16"],
              "rsm_filter_params": ["sor_mean_k", "sor_std_mul", "normal_radius", "cam_center"]}
    body = "".join('printf("%s %%zu\\n", sizeof(%s));\n' % (t, t) + "".join('printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (t, f, t, f) for f in fs)
                   for t, fs in fields.items())
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rsm.h"\nint main(void) {\n%sprintf("rsm_point16 %%zu\\n", sizeof(rsm_point16));\nreturn 0; }\n' % body)
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-I" + os.path.dirname(HEADER), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    mirror = {"rsm_pair_in": _lib.PairIn, "rsm_pair_out": _lib.PairOut, "rsm_filter_params": _lib.FilterParams}
    for t, fs in fields.items():
        assert int(got[t]) == C.sizeof(mirror[t]), t
        for f in fs:
            assert int(got["%s.%s" % (t, f)]) == getattr(mirror[t], f).offset, (t, f)
    assert int(got["rsm_point16"]) == 16


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from reconstruction_amd import Context, RsmError
    with pytest.raises(RsmError):
        Context(0)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "reconstruction_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                # comments may mention the oracle; nothing may import, include, link or load it
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", txt), f
                assert "liborc" not in txt and "orc_match_pair" not in txt, f


def test_header_is_plain_c():
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-x", "c", HEADER], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/reconstruction"), reason="reference headers only exist in the build container")
@pytest.mark.parametrize("pcl", [False, True])
def test_cpp_adapter_compiles_against_the_reference_headers(tmp_path, pcl):
    """The cv::Mat-facing shim (include/CStereoMatchingMI355.hpp) against the reference's own vendored headers, with and
    without IS_PCL (SharedInclude.h defines it; the InsertPoint / filter calls are behind it, CStereoMatching.cpp:30,750).
    The marshalling itself is plain C++ and is RUN by tests/test_gpu_cpp_adapter.py."""
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference checkout exists only in the build container")
    src = tmp_path / "adapter_check.cpp"
    src.write_text('#define __declspec(x)\n#define _Longlong long long\n#include "SharedInclude.h"\n'
                   + ('' if pcl else '#undef IS_PCL\n') +
                   '#include "CStereoMatching.h"\n#include "CStereoMatchingMI355.hpp"\n'
                   'bool f(CStereoMatching &sm) { RsmStereoMI355 g(0); return g.MatchPair(sm, 0) && g.LastStatus() == 0; }\n'
                   # the whole loop as INTEGRATION.md 2 writes it: a member function, so the lambda may call the private Rectify
                   'void CStereoMatching::MatchAllLayer() {\n'
                   '    static RsmStereoMI355 gpu(0, 3);\n'
                   '    RsmCvTraits::rectify() = [](CStereoMatching &s, int CamPair) { s.Rectify(CamPair, s.Q); };\n'
                   '    std::vector<int> status(m_data->m_CampairNum);\n'
                   '    if (gpu.MatchAll(*this, m_data->m_CampairNum, status.data()) != m_data->m_CampairNum) printf("rsm: %s\\n", gpu.LastError());\n'
                   '}\n'
                   # the several-GPU form (every visible device, three pairs in flight on each) and the loop with the filter on the GPU
                   'int g(CStereoMatching &sm, int n) {\n'
                   '    static RsmStereoMI355 gpu(RsmStereoMI355::AllDevices(), 3);\n'
                   '    RsmCvTraits::filtered_sink() = [](CStereoMatching &, int, const rsm_point16 *, const float *, int64_t, int64_t) {};\n'
                   '    return gpu.MatchAll(sm, n) + gpu.MatchAllFiltered(sm, n);\n'
                   '}\n')
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-w", "-I/root/reference/include",
                        "-I/root/reference/reconstruction", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_mock_adapter_program_builds_and_links(tmp_path):
    """tests/cpp/mock_adapter.cpp (the traits-based adapter with mock types) compiles with plain g++ and links against
    the C-ABI library; it is executed on the GPU box."""
    from reconstruction_amd import _lib
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "mock_adapter.cpp"), "-o", str(tmp_path / "mock_adapter"),
                        "-L" + os.path.dirname(_lib.LIB_PATH), "-lrsm_mi355", "-Wl,-rpath-link,/opt/rocm/lib",
                        "-Wl,--allow-shlib-undefined", "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_every_option_of_rsm_set_option_is_documented_in_the_header():
    """include/rsm.h lists the tuning knobs; a name rsm_set_option accepts and the header does not mention is a knob nobody
    can find."""
    import re
    src = open(os.path.join(ROOT, "reconstruction_amd", "csrc", "rsm_api.hip")).read()
    names = sorted(set(re.findall(r'!strcmp\(name, "([a-z_0-9]+)"\)', src)))
    assert len(names) >= 20
    hdr = open(os.path.join(ROOT, "include", "rsm.h")).read()
    missing = [n for n in names if '"%s"' % n not in hdr]
    assert not missing, missing
