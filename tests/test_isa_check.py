"""CPU tests of the build-time ISA audit (lush_nerf_amd/isa_check.py): each rule on a minimal instruction sequence in
llvm-objdump's format, the path-sensitive walk (loop back-edge, both sides of a branch), the matrix-pipe ageing, and the
product library itself."""
import os
import re

import pytest

from lush_nerf_amd import isa_check as C, lib


def _kernel(body: str) -> str:
    """Lines `op operands` -> objdump text with addresses (4 bytes per instruction; branch targets as `@label`)."""
    lines = [l.strip() for l in body.strip().split("\n") if l.strip()]
    labels, insts = {}, []
    for l in lines:
        if l.endswith(":"):
            labels[l[:-1]] = len(insts)
        else:
            insts.append(l)
    out = ["0000000000001000 <k>:"]
    for n, l in enumerate(insts):
        m = re.search(r"@(\w+)", l)
        if m:       # simm16 = (target - (addr + 4)) / 4
            simm = labels[m.group(1)] - (n + 1)
            l = l.replace("@" + m.group(1), str(simm & 0xFFFF))
        out.append(f"\t{l}    // {0x1000 + 4 * n:012X}: 00000000")
    return "\n".join(out)


def _check(body: str, name="k"):
    ks = C.parse_kernels(_kernel(body))
    return C.check_kernel(name, ks["k"])


def test_r1_valu_written_sgpr_into_vmem_base():
    bad = """
        v_readlane_b32 s18, v254, 8
        v_readlane_b32 s19, v254, 9
        global_load_dwordx4 v[2:5], v2, s[18:19]
        s_waitcnt vmcnt(0)
        s_endpgm
    """
    f = _check(bad)
    assert len(f) == 2 and all(x.startswith("R1") for x in f), f
    ok = bad.replace("global_load", "s_nop 4\n        global_load")
    assert _check(ok) == []
    # an SALU copy in between is no cure by itself: the VALU write is still inside its window for the VMEM that reads the SAME register
    assert any(x.startswith("R1") for x in _check(bad.replace("global_load", "s_mov_b32 s4, s18\n        global_load")))
    # ... but a VMEM that reads only the SALU copy is fine
    assert _check("""
        v_readfirstlane_b32 s18, v3
        s_mov_b32 s4, s18
        s_mov_b32 s5, 0
        global_load_dword v1, v2, s[4:5]
        s_waitcnt vmcnt(0)
        s_endpgm
    """) == []


def test_r2_store_data_overwritten_too_early():
    bad = """
        global_store_dwordx4 v14, v[10:13], s[14:15] nt
        v_mov_b32_e32 v11, 0
        s_endpgm
    """
    f = _check(bad)
    assert len(f) == 1 and f[0].startswith("R2"), f
    assert _check(bad.replace("v_mov", "s_nop 1\n        v_mov")) == []
    # a DS read into the data registers returns long after the store has read them: not a hazard
    assert _check(bad.replace("v_mov_b32_e32 v11, 0", "ds_read_b128 v[10:13], v9")) == []


def test_r3_valu_into_mfma_and_packed_fp32():
    bad = """
        v_mul_f32_e32 v114, v114, v116
        v_mfma_f32_32x32x16_f16 v[100:115], v[0:3], v[4:7], v[100:115]
        s_endpgm
    """
    assert [x[:2] for x in _check(bad)] == ["R3"]
    assert _check(bad.replace("v_mfma", "s_nop 1\n        v_mfma")) == []
    pk = bad.replace("v_mul_f32_e32 v114, v114, v116", "v_pk_mul_f32 v[114:115], v[114:115], v[116:117]")
    assert [x[:2] for x in _check(pk.replace("v_mfma", "s_nop 1\n        v_mfma"))] == ["R3"]          # 2 is not enough for packed fp32
    assert _check(pk.replace("v_mfma", f"s_nop {C.PK_F32_TO_MFMA - 1}\n        v_mfma")) == []


def test_r4_mfma_result_and_matrix_pipe_ageing():
    # the partial-liveness case of round 4: an asm output allocated in the dead tail of an in-flight accumulator block
    bad = """
        v_mfma_f32_32x32x16_f16 v[0:15], v[74:77], v[198:201], v[0:15]
        v_cvt_pk_f16_f32 v173, v72, v73
        v_pk_sub_u16 v1, 1, v172 op_sel_hi:[0,1] clamp
        s_endpgm
    """
    f = _check(bad)
    assert len(f) == 1 and f[0].startswith("R4") and "1 wait state" in f[0], f
    assert _check(bad.replace("v_pk_sub", "s_nop 9\n        v_pk_sub")) == []
    # an MFMA in between issues only when the matrix pipe is free: it ages the producer by its 8 passes
    aged = """
        v_mfma_f32_32x32x16_f16 v[0:15], v[74:77], v[198:201], v[0:15]
        v_mfma_f32_32x32x16_f16 v[16:31], v[74:77], v[198:201], v[16:31]
        v_mfma_f32_32x32x16_f16 v[32:47], v[74:77], v[198:201], v[32:47]
        v_cvt_pk_f16_f32 v173, v0, v1
        s_endpgm
    """
    assert _check(aged) == []
    # the accumulate chain itself (the next MFMA takes D whole as C) is not a finding
    assert _check("""
        v_mfma_f32_32x32x16_f16 v[0:15], v[74:77], v[198:201], v[0:15]
        v_mfma_f32_32x32x16_f16 v[0:15], v[78:81], v[198:201], v[0:15]
        s_endpgm
    """) == []


R5_NAME = "k_mlp_wide_bwd_kernel"


def _check5(body):
    ks = C.parse_kernels(_kernel(body))
    return C.check_kernel(R5_NAME, ks["k"])


def test_r5_outstanding_load_across_the_back_edge():
    # a prefetch at the end of the loop body, 2 stores behind it, and a counted wait at the head of the next iteration
    loop = """
        global_load_dwordx4 v[2:5], v2, s[10:11]
        s_waitcnt vmcnt(0)
    head:
        s_waitcnt vmcnt(%d)
        v_add_f32_e32 v9, v2, v3
        global_load_dwordx4 v[2:5], v2, s[10:11]
        global_store_dwordx4 v14, v[20:23], s[50:51] nt
        global_store_dwordx4 v14, v[20:23], s[50:51] nt
        s_cbranch_scc1 @head
        s_waitcnt vmcnt(0)
        s_endpgm
    """
    assert _check5(loop % 2) == []                       # all but the 2 youngest (the stores): the load has landed
    f = _check5(loop % 3)                                # one too many: the load may still be in flight
    assert f and all(x.startswith("R5") for x in f), f
    # the form round 3 shipped: two waits selected by complementary conditions -- correct at run time, but on the path that
    # takes neither no wait retires the load; the audit asks for code that is right on every path
    two = """
        global_load_dwordx4 v[2:5], v2, s[10:11]
        s_cbranch_scc0 @a
        s_waitcnt vmcnt(32)
    a:
        s_cbranch_vccnz @b
        s_waitcnt vmcnt(0)
    b:
        v_add_f32_e32 v9, v2, v3
        s_endpgm
    """
    assert any(x.startswith("R5") for x in _check5(two))
    # R5 is applied to the kernels listed in R5_KERNELS only
    ks = C.parse_kernels(_kernel(two))
    assert C.check_kernel("some_other_kernel", ks["k"]) == []


def test_every_asm_register_load_is_in_a_kernel_r5_covers():
    """R5_KERNELS must name the kernels of every source file that hides a VMEM load into registers in an asm statement."""
    files = {}
    for f in sorted(os.listdir(lib.CSRC)):
        src = open(os.path.join(lib.CSRC, f)).read()
        if re.search(r'asm volatile\(\s*"(?:[^"]*\\n\\t)?(global|buffer|flat|scratch)_load_(?!lds)(?![a-z_0-9]*lds)', src):
            files[f] = re.findall(r"__global__[^\n]*?void\s+(\w+)\s*\(", src) + re.findall(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", src)
    assert set(files) == {"lush_mlp_wide_bwd.hip"}, files
    for f, kernels in files.items():
        assert kernels, f
        for k in set(kernels):
            assert any(n in k for n in C.R5_KERNELS), (f, k)


def test_the_product_library_passes_the_audit():
    lib.build()
    n, found = C.check_shared_object(lib.SO_PATH)
    assert n >= 80 and found == [], found[:5]
