"""GPU: the C-ABI from a host that is not Python and links no torch -- examples/abi_host.cpp is compiled with hipcc against
include/icka_hip.h and icka_amd/libicka_hip.so and run as its own process: an nn.Linear + fused bias / residual / LayerNorm on
hipMalloc'ed buffers and a stream of its own, checked against double-precision host loops (the boundary is
`extern "C"`, plain pointers and sizes)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_binds_the_c_abi_without_torch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "abi_host")
    lib_dir = os.path.join(ROOT, "icka_amd")
    build = subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "abi_host.cpp"), "-L" + lib_dir, "-licka_hip",
                            "-Wl,-rpath," + lib_dir, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert build.returncode == 0, build.stdout.decode("utf-8", "replace")[-3000:]
    ldd = subprocess.run(["ldd", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    assert "libicka_hip.so" in ldd and "torch" not in ldd and "python" not in ldd.lower(), ldd
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = run.stdout.decode("utf-8", "replace")
    print("\n" + out.strip())
    assert run.returncode == 0, out[-2000:]
    assert "max abs err" in out
