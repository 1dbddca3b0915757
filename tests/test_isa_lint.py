"""The ISA lint for asynchronous inline-asm loads (etch_amd/isa_lint.py): the checker itself on hand-written snippets, and the shipped kernels that
issue loads from inline asm with hand-counted waits (VERDICT r04 item 2 / ADVICE r04: the hazard class behind profiles/r04_x32_cin32_miscompile.txt
must be absent from what ships).  Cross-compiles for gfx950: no GPU needed."""
import os

import pytest

from etch_amd import isa_lint as L

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "etch_amd", "csrc")


def _kernel(body):
    return "_Zk:\n" + body + "\n\ts_endpgm\n\t.end_amdhsa_kernel\n"


def test_lint_flags_a_copy_between_an_asm_load_and_its_wait():
    bad = _kernel("\t;;#ASMSTART\n\tds_read_b128 v[2:5], v1\n\t;;#ASMEND\n\tv_mov_b32_e32 v22, v5\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND")
    f = L.lint_asm(bad, only_asm_loads=True)
    assert len(f) == 1 and f[0]["reg"] == "v5" and f[0]["load_in_asm"]
    good = _kernel("\t;;#ASMSTART\n\tds_read_b128 v[2:5], v1\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v22, v5")
    assert L.lint_asm(good, only_asm_loads=True) == []


def test_lint_counter_model_is_in_order_per_kind_only():
    # all but the youngest: the older LDS read is complete after lgkmcnt(1) ...
    ok = _kernel("\t;;#ASMSTART\n\tds_read_b128 v[2:5], v1\n\t;;#ASMEND\n\t;;#ASMSTART\n\tds_read_b128 v[6:9], v1\n\t;;#ASMEND\n\ts_waitcnt lgkmcnt(1)\n\tv_add_f32_e32 v0, v2, v3")
    assert L.lint_asm(ok, only_asm_loads=True) == []
    # ... but a plain load is NOT complete after vmcnt(1) when the younger operation is an LDS-direct load (out of order between the kinds)
    mixed = _kernel("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[2:5], v[10:11], off\n\t;;#ASMEND\n\tglobal_load_lds_dwordx4 v[12:13], off\n\ts_waitcnt vmcnt(1)\n\tv_add_f32_e32 v0, v2, v3")
    assert len(L.lint_asm(mixed, only_asm_loads=True)) == 2
    # and a loop: a load that crosses the back edge is seen by the next iteration
    loop = _kernel(".LBB0_1:\n\tv_add_f32_e32 v0, v2, v3\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[2:5], v[10:11], off\n\t;;#ASMEND\n\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(0)")
    f = L.lint_asm(loop, only_asm_loads=True)
    assert {(x["instr"].split()[0], x["reg"]) for x in f} >= {("v_add_f32_e32", "v2"), ("v_add_f32_e32", "v3")}      # (+ the reload into registers still in flight)


@pytest.mark.parametrize("src", ["so3conv_y.hip", "so3conv_x.hip", "so3conv.hip"])
def test_shipped_kernels_never_touch_an_asm_load_before_its_wait(src):
    text = L.compile_to_asm(os.path.join(CSRC, src))
    n_asm_loads = sum(1 for ln in text.split("\n") if ln.strip().startswith(("global_load_dwordx4", "ds_read_b128")))
    assert n_asm_loads > 0
    f = L.lint_asm(text, only_asm_loads=True)
    assert f == [], f[:5]


def test_lint_flags_an_asm_valu_read_of_a_fresh_mfma_result():
    """The hazard behind the first build of the fp16 attention layer: `v_max3_f32` in inline asm straight after the MFMA that writes its operand."""
    mfma = "\tv_mfma_f32_16x16x32_f16 v[4:7], v[8:11], v[12:15], 0\n"
    bad = _kernel(mfma + "\t;;#ASMSTART\n\tv_max3_f32 v0, v4, v5, v6\n\t;;#ASMEND")
    f = L.lint_mfma_asm_reads(bad)
    assert len(f) == 3 and {x["reg"] for x in f} == {"v4", "v5", "v6"} and f[0]["distance"] == 1
    # the destination of the asm instruction is not a read
    assert L.lint_mfma_asm_reads(_kernel(mfma + "\t;;#ASMSTART\n\tv_max3_f32 v4, v1, v2, v3\n\t;;#ASMEND")) == []
    # a compiler-emitted read first (it gets the wait states; older MFMAs are complete too) ...
    ok1 = _kernel("\tv_mfma_f32_16x16x32_f16 v[20:23], v[8:11], v[12:15], 0\n" + mfma + "\tv_mul_f32_e32 v1, v2, v7\n\t;;#ASMSTART\n\tv_max3_f32 v0, v4, v20, v6\n\t;;#ASMEND")
    assert L.lint_mfma_asm_reads(ok1) == []
    # ... but not when it read an OLDER result only
    bad2 = _kernel("\tv_mfma_f32_16x16x32_f16 v[20:23], v[8:11], v[12:15], 0\n" + mfma + "\tv_mul_f32_e32 v1, v2, v21\n\t;;#ASMSTART\n\tv_max3_f32 v0, v4, v20, v6\n\t;;#ASMEND")
    assert {x["reg"] for x in L.lint_mfma_asm_reads(bad2)} == {"v4", "v6"}
    # the guard block of ml_attention: >= 16 wait states of s_nop in inline asm
    ok2 = _kernel(mfma + "\t;;#ASMSTART\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3\n\t;;#ASMEND\n\t;;#ASMSTART\n\tv_max3_f32 v0, v4, v5, v6\n\t;;#ASMEND")
    assert L.lint_mfma_asm_reads(ok2) == []
    # write-after-read: an asm result landing in a register that an MFMA issued a moment ago still reads as its accumulator input
    war = _kernel("\tv_mfma_f32_16x16x32_f16 v[16:19], v[8:11], v[12:15], v[4:7]\n\t;;#ASMSTART\n\tv_fma_mix_f32 v5, v1, -1.0, v2 op_sel_hi:[1,0,0]\n\t;;#ASMEND")
    f = L.lint_mfma_asm_reads(war)
    assert len(f) == 1 and f[0]["reg"] == "v5" and f[0].get("kind") == "write-after-read"
    # in place on the accumulator of an in-place MFMA chain is a read of a fresh result instead (caught above); a compiler-emitted write first is fine
    ok3 = _kernel("\tv_mfma_f32_16x16x32_f16 v[16:19], v[8:11], v[12:15], v[4:7]\n\tv_mov_b32_e32 v4, v1\n\t;;#ASMSTART\n\tv_fma_mix_f32 v5, v1, -1.0, v2 op_sel_hi:[1,0,0]\n\t;;#ASMEND")
    assert L.lint_mfma_asm_reads(ok3) == []


@pytest.mark.parametrize("src", ["mhsa_layer.hip", "so3conv_y.hip", "so3conv_ws.hip", "fused_dense.hip", "so3conv.hip"])
def test_shipped_kernels_never_read_a_fresh_mfma_result_from_inline_asm(src):
    """Every kernel source that mixes MFMAs with inline-asm VALU instructions (v_fma_mix_f32, v_max3_f32)."""
    text = L.compile_to_asm(os.path.join(CSRC, src))
    assert "v_mfma" in text
    f = L.lint_mfma_asm_reads(text)
    assert f == [], f[:5]
