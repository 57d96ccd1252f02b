"""include/needle_hip.h is a C header: a C99 translation unit (tests/c/abi_smoke.c) compiles against it with
-pedantic, links with libneedle_hip.so and exercises the device-free part of the ABI.  CPU only."""
import os
import subprocess

from conftest import ROOT


def test_header_compiles_as_c_and_links(tmp_path):
    from needle_amd import build
    lib = build.build()
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(lib)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L", libdir, "-lneedle_hip", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c abi ok" in r.stdout


def build_example(tmp_path):
    from needle_amd import build
    lib = build.build()
    exe = str(tmp_path / "scan_rows")
    libdir = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "scan_rows.c"), "-L", libdir, "-lneedle_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_example_builds(tmp_path):
    assert os.path.exists(build_example(tmp_path))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    """examples/scan_rows.c end to end on the device: a C host, device buffers from hipMalloc, the three result kinds."""
    r = subprocess.run([build_example(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert 'row 0 "order 66 shipped": found=1 first=(6,8) all=(6,8)' in out
    assert 'row 1 "no digits here": found=0 first=(-1,-1) all=' in out
    assert 'row 3 "a1b22c333": found=1 first=(1,2) all=(1,2)(3,5)(6,9)' in out
    assert 'row 4 "2024-01-31": found=1 first=(0,4) all=(0,4)(5,7)(8,10)' in out
    assert "more=0" in out
    assert "csr total=7 more=0: (6,8) (1,2) (3,5) (6,9) (0,4) (5,7) (8,10)" in out
    assert "packed: (6,8) -" in out and out.split("packed:")[1].split("\n")[0].split()[3:] == ["(1,2)", "(0,4)"]
    assert 'tuning info: 1 bytes, header "name\tdefault\tcurrent\tscope\teffect"' in out
