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
    assert lib.icka_abi_version() == _lib.ABI_VERSION == 6
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
          ' sizeof(icka_slab_reduction), __builtin_offsetof(icka_slab_reduction, out));' \
          ' printf(" %zu %llu", __builtin_offsetof(icka_gemm_desc, tune), (unsigned long long)(ICKA_TUNE_RING(4) | ICKA_TUNE_TILE_N(96)' \
          ' | ICKA_TUNE_WIDE_TILES(0) | ICKA_TUNE_DIRECT_EPILOGUE(1) | ICKA_TUNE_WARP_SPECIALIZED(2) | ICKA_TUNE_W3_GRID(4)' \
          ' | ICKA_TUNE_BIG_TILES(1)));return 0;}\n'
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
    assert int(out[5]) == GemmDesc.tune.offset
    # the header's ICKA_TUNE_* macros and kernels.gemm_tune build the same word
    from icka_amd import kernels
    assert int(out[6]) == kernels.gemm_tune(ring=4, tile_n=96, wide_tiles=False, direct_epilogue=True, warp_specialized=2, w3_grid=4,
                                            big_tiles=1)


def test_no_tuning_setters_in_the_exported_abi():
    """VERDICT r04 item 7 / SURVEY.md section 8b (re-entrant launchers, no hidden state): the shared object exports NO ``*_set_*``
    symbol other than the documented process-wide controls (dropout nonce registration, the CU reservation beside RCCL); what
    varies per launch is an argument (icka_gemm_desc.tune, flags) or an ICKA_TUNE_* environment variable read once at load."""
    import subprocess
    from icka_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("icka_")})
    assert len(exported) > 100
    setters = [n for n in exported if "_set_" in n]
    assert setters == ["icka_lstm_set_reserved_cus", "icka_set_dropout_nonce"], setters
    assert not [n for n in exported if n.startswith("icka_diag_")]          # diagnostic entry points exist in diagnostic builds only
    assert sorted(_header_functions()) == exported                            # nothing exported that the header does not declare


def test_no_cpu_path():
    """Product launchers refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from icka_amd import kernels
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        kernels.gemm(kernels.GEMM_NT, a, a, torch.zeros(8, 8, dtype=torch.bfloat16))
