"""CPU: the C-ABI library loads and exports exactly the symbols include/icka_hip.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "icka_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icka_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from icka_amd import _lib
    assert _header_functions() == sorted(_lib.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    from icka_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for name in _header_functions():
        assert hasattr(lib, name), name
    assert lib.icka_abi_version() == _lib.ABI_VERSION == 5
    assert lib.icka_build_arch() == b"gfx950"
    assert lib.icka_ln_bwd_workspace_floats(768) == 1024 * 4 * 768


def test_gemm_desc_layout_matches_c_struct():
    """sizeof(icka_gemm_desc) as the C compiler lays it out == ctypes mirror."""
    import ctypes
    import subprocess
    import tempfile
    from icka_amd._lib import GemmDesc, SlabReduction
    src = '#include <stdio.h>\n#include "icka_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu", sizeof(icka_gemm_desc),' \
          ' __builtin_offsetof(icka_gemm_desc, bias), __builtin_offsetof(icka_gemm_desc, epilogue),' \
          ' sizeof(icka_slab_reduction), __builtin_offsetof(icka_slab_reduction, out));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")],
                       check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(GemmDesc)
    assert int(out[1]) == GemmDesc.bias.offset
    assert int(out[2]) == GemmDesc.epilogue.offset
    assert int(out[3]) == ctypes.sizeof(SlabReduction)
    assert int(out[4]) == SlabReduction.out.offset


def test_no_cpu_path():
    """Product launchers refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from icka_amd import kernels
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        kernels.gemm(kernels.GEMM_NT, a, a, torch.zeros(8, 8, dtype=torch.bfloat16))
