"""tools/lint_asm_waits.py: the build-time check that no instruction touches the destination of an inline-asm LDS read before the
counted s_waitcnt that covers it (the fault it was written for: a register-allocator copy of half a V fragment ahead of its wait in
attn_bwd3_dq_kernel's ragged-tile path -- rare huge / NaN dQ rows at ViT size, never in the small parity cases)."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lint_asm_waits", os.path.join(ROOT, "tools", "lint_asm_waits.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

HEAD = "_Z6kernelv:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"


def asm(*ins):
    return "".join(f"\t;;#ASMSTART\n\t{i}\n\t;;#ASMEND\n" for i in ins)


def run(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text(HEAD + body + TAIL)
    funcs = lint.parse_functions(str(f))
    (name, items), = funcs.items()
    return lint.lint_function(name, items)


def test_copy_ahead_of_the_wait_is_reported(tmp_path):
    body = (asm("ds_read_b128 v[78:81], v201 offset:0x2000", "ds_read_b128 v[82:85], v201 offset:0x3000")
            + "\tv_mov_b64_e32 v[186:187], v[80:81]\n" + asm("s_waitcnt lgkmcnt(0)"))
    bad = run(tmp_path, body)
    assert len(bad) == 1 and bad[0][4] == "reads" and "v_mov_b64" in bad[0][3]


def test_counted_wait_retires_in_order(tmp_path):
    # two reads, lgkmcnt(1): the first is complete, the second is not
    body = (asm("ds_read_b128 v[0:3], v9 offset:0", "ds_read_b128 v[4:7], v9 offset:0x1000", "s_waitcnt lgkmcnt(1)")
            + "\tv_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[12:15], v[16:31]\n" + asm("s_waitcnt lgkmcnt(0)")
            + "\tv_mfma_f32_32x32x16_bf16 v[16:31], v[4:7], v[12:15], v[16:31]\n")
    assert run(tmp_path, body) == []
    body = body.replace("v[0:3], v[12:15]", "v[4:7], v[12:15]", 1)
    assert len(run(tmp_path, body)) == 1


def test_paths_through_branches_are_followed(tmp_path):
    # the wait sits on one side of a branch only
    body = (asm("ds_read_b128 v[0:3], v9 offset:0") + "\ts_cbranch_vccz .LBB0_2\n" + asm("s_waitcnt lgkmcnt(0)")
            + ".LBB0_2:\n\tv_add_f32_e32 v8, v0, v1\n")
    bad = run(tmp_path, body)
    assert len(bad) == 1 and bad[0][4] == "reads"


def test_vmcnt_only_wait_does_not_retire_lds_reads(tmp_path):
    body = asm("ds_read_b128 v[0:3], v9 offset:0", "s_waitcnt vmcnt(0)") + "\tv_add_f32_e32 v8, v0, v1\n"
    assert len(run(tmp_path, body)) == 1


def test_built_kernels_are_clean():
    """The ISA `make` produced for the sources that use the idiom (csrc/build/*.s, written by build())."""
    bdir = os.path.join(ROOT, "item_alignment_amd", "csrc", "build")
    files = [os.path.join(bdir, f) for f in ("gemm.s", "attention.s")]
    if not all(os.path.exists(f) for f in files):
        pytest.skip("csrc/build/*.s not built here (make -C item_alignment_amd/csrc)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_asm_waits.py"), *files], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "attn_bwd3_dq_kernel" in r.stdout and "attn_fwd3_kernel" in r.stdout
