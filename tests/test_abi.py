"""The C-ABI library builds, loads and exports every symbol include/pwicp.h declares (no compute here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pwicp.h")).read()
    return sorted(set(re.findall(r"PWICP_API\s+[\w\s\*]+?\b(pwicp_\w+)\s*\(", txt)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ("pwicp_create", "pwicp_nn_search", "pwicp_percentile_dist", "pwicp_patch_normals",
                 "pwicp_select_patches", "pwicp_p2p_icp", "pwicp_trans_para_vcm", "pwicp_pair_create",
                 "pwicp_pair_run", "pwicp_overlap_ratio"):
        assert must in syms
    assert len(syms) >= 18


def test_library_exports_every_declared_symbol():
    import pwicp_amd
    lib = ctypes.CDLL(pwicp_amd.lib_path())
    for s in declared_symbols():
        assert hasattr(lib, s), "libpwicp.so does not export %s" % s
    assert b"gfx950" in pwicp_amd.load_library().pwicp_version()


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "pwicp.h")).read()
    assert txt.count("R.cpp:") + txt.count("C.cpp:") + txt.count("S.cpp:") >= 15


def test_no_device_fails_loudly():
    """Without a GPU pwicp_create must fail with NO_DEVICE — there is no CPU fallback in the product."""
    import pwicp_amd
    if pwicp_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pwicp_amd.PwicpError) as e:
        pwicp_amd.Context(0)
    assert e.value.code == -1


def test_product_does_not_reference_oracle():
    """The product sources never include, link or import anything under oracle/."""
    pkg = os.path.join(ROOT, "piecewise-icp_amd")
    for d, _, files in os.walk(pkg):
        if os.sep + "build" in d:
            continue
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".py", "Makefile")):
                t = open(os.path.join(d, f), errors="replace").read()
                assert "pwicp_oracle" not in t and "_oracle" not in t and "oracle/" not in t, os.path.join(d, f)


def _build_facade_check(tmp_path):
    import subprocess
    import pwicp_amd
    exe = str(tmp_path / "facade_check")
    libdir = os.path.dirname(pwicp_amd.lib_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "facade_compile_check.cpp"), "-L" + libdir, "-lpwicp",
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_cpp_facade_compiles_and_links(tmp_path):
    """include/pwicp/Registration.h (the reference's function names over the C ABI) instantiates and links."""
    import subprocess
    out = subprocess.run([_build_facade_check(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "instantiated: 9" in out.stdout


@pytest.mark.gpu
def test_cpp_facade_runs_a_registration(tmp_path):
    import subprocess
    out = subprocess.run([_build_facade_check(tmp_path), "run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "FACADE_OK" in out.stdout, out.stdout + out.stderr


def test_header_is_plain_c99(tmp_path):
    """include/pwicp.h compiles as strict C99 (-pedantic), links against libpwicp.so and the 384-byte record has the
    size the all-gather assumes; without a GPU pwicp_create reports PWICP_E_NO_DEVICE (-1), never a fallback."""
    import subprocess
    import pwicp_amd
    exe = str(tmp_path / "c_abi_check")
    libdir = os.path.dirname(pwicp_amd.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_check.c"), "-L" + libdir, "-lpwicp", "-Wl,-rpath," + libdir,
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "record 384 B" in out.stdout
    assert ("-> -1" in out.stdout) or ("-> 0" in out.stdout)


def _build_facade_pcl_check(tmp_path):
    import subprocess
    import pwicp_amd
    exe = str(tmp_path / "facade_pcl_check")
    libdir = os.path.dirname(pwicp_amd.lib_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "shim"),
                           os.path.join(ROOT, "tests", "facade_pcl_check.cpp"), "-L" + libdir, "-lpwicp", "-Wl,-rpath," + libdir,
                           "-o", exe])
    return exe


def test_exact_reference_signatures_compile_against_the_type_shim(tmp_path):
    """The `#if PWICP_HAVE_PCL` block of include/pwicp/Registration.h — the reference's exact signatures of Registration.h,
    Segmentation.h and CommonFunc.h (SURVEY 8b "C++ API to keep") — compiled against tests/shim (PointXYZ 16 B, PointNormal
    48 B, PointCloud<T>::Ptr, Matrix4f, MatrixXd) and linked against libpwicp.so."""
    import subprocess
    out = subprocess.run([_build_facade_pcl_check(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "exact signatures compiled: 20" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_exact_reference_signatures_run(tmp_path):
    """Every exact-signature function called once on the GPU; Piecewise_ICP equals the loop of R.cpp:668-700 written with
    PwICP_singleIteration, bit for bit."""
    import subprocess
    exe = _build_facade_pcl_check(tmp_path)
    out = subprocess.run([exe, "run", os.path.join(ROOT, "tests", "golden", "reference_results"), str(tmp_path)],
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0 and "FACADE_PCL_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
