"""The C-ABI library loads on a CPU-only box and exports every symbol include/itemalign.h declares; the ctypes
binding lists exactly those symbols.  (No compute calls here: those need the GPU.)"""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "itemalign.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ia_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    fns = header_functions()
    for must in ("ia_gemm_bf16", "ia_attn_fwd", "ia_attn_bwd", "ia_ln_fwd", "ia_ln_bwd", "ia_embed_ln_fwd", "ia_embed_ln_bwd",
                 "ia_pair_head_ce_fwd", "ia_pair_head_ce_bwd", "ia_adamw_flat", "ia_layer_fwd", "ia_layer_bwd", "ia_strerror"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from item_alignment_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/itemalign.h but not exported"


def test_binding_covers_the_header():
    from item_alignment_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    lib = _lib.load()
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'itemalign.h')).read()
    assert lib.ia_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define IA_ABI_VERSION (\d+)', header).group(1))
    assert lib.ia_strerror(-1).decode().startswith("invalid argument")
    assert lib.ia_strerror(0).decode() == "ok"


def test_error_codes_without_touching_the_gpu():
    """argument validation happens before any launch: null pointers / bad shapes -> IA_ERR_ARG."""
    from item_alignment_amd import _lib
    lib = _lib.load()
    assert lib.ia_gemm_bf16(None, 0, 8, None, 0, 8, None, 0, 8, 8, 8, 8, 0, None, None, 0, None, 0, None, 0, None) == -1
    assert lib.ia_ln_fwd(None, None, None, None, None, None, None, None, None, 4, 64, 1e-5, 0.0, 0, 0, None) == -1
    assert lib.ia_gemm_workspace_bytes(1024, 1024, 32640, 1) > 0
    assert lib.ia_gemm_workspace_bytes(1024, 1024, 32640, 0) == 0
    assert lib.ia_ln_bwd_workspace_bytes(1000, 1024) == 250 * 3 * 1024 * 4


def test_library_has_no_undefined_kernel_stubs():
    """A shared library links with undefined symbols: a kernel whose host-side launch stub hipcc failed to emit (seen when a lambda
    call is written directly as an argument of a builtin inside a template kernel) would only fail at the first launch."""
    import os
    import subprocess
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "item_alignment_amd", "libitemalign_hip.so")
    out = subprocess.run(["nm", "-C", "-u", lib], capture_output=True, text=True, check=True).stdout
    bad = [ln for ln in out.splitlines() if "__device_stub__" in ln]
    assert not bad, bad
