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
