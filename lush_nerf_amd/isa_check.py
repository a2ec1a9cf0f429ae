"""Build-time audit of the gfx950 code objects in liblush_march.so: the hazards hipcc does not cover for `asm volatile`.

hipcc schedules an asm statement as one opaque instruction: it neither counts the memory operations inside nor pads the
wait states the ISA asks for between an instruction inside the string and compiler code around it.  Round 3 met three of
these the hard way ("Hazards"): a GPU memory-access fault and two silent miscomputations.  This
module disassembles what was actually built (llvm-objdump) and checks every kernel against the rules, so a build in
which the compiler's register allocation or scheduling happens to create one of them is refused HERE, in the build
container, instead of being found on the GPU one wrong result at a time.  lib.build() runs it.

Rules (wait state = one issue slot; `s_nop N` = N + 1; an MFMA issues only when the matrix pipe is free, i.e. `passes` wait
states after the previous MFMA, so MFMAs in between age a producer by their pass count):
  R1  VALU writes an SGPR (v_readlane_b32 = the compiler's scalar-spill reload, v_readfirstlane_b32, v_cmp_*_e64 sdst, carry
      outs)  ->  a VMEM instruction reads that SGPR (scalar base / offset / descriptor): 5 wait states.
      (round 3's fault: spill reload `v_readlane_b32 s18 / s19` directly in front of the asm `global_load_dwordx4 v, v, s[18:19]`)
  R2  VMEM store of more than 8 bytes per lane  ->  write of its data VGPRs: 2 wait states (gfx940+).
  R3  VALU writes a VGPR  ->  MFMA reads it as A, B or C: 2 wait states; packed fp32 (v_pk_{mul,add,fma}_f32): PK_F32_TO_MFMA.
  R4  MFMA writes D  ->  a VALU / DS / VMEM instruction reads or overwrites those VGPRs: passes + 3 wait states (11 for
      32x32x16: what hipcc pads for its own code on this toolchain).
  R5  A VGPR that an outstanding VMEM load will write is read or written before an `s_waitcnt vmcnt(N)` has retired that
      load (vmcnt retires in issue order; stores and LDS-DMAs count) -- this validates every hand-counted vmcnt(N) behind
      an asm load, including across the loop back-edge.
Control flow: every path is followed (a worklist over (instruction, state), both sides of every conditional branch), so a
producer at the end of a loop body meets the consumer at its head.
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile
from typing import Dict, List, Optional, Tuple

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
# R5 follows every outstanding VMEM operation along every path, which is only tractable (and only needed) where a register
# load is hidden from the compiler's own s_waitcnt bookkeeping: the kernels of the source files that hold an asm VMEM load
# into registers (tests/test_cpu_host.py checks this list against the sources).
R5_KERNELS = ("mlp_wide_bwd_kernel",)
PK_F32_TO_MFMA = 4          # tools/micro/hazards.hip P2 measures what the hardware needs; see DESIGN.md section 4
_REG = re.compile(r"\b([vsa])(?:\[(\d+):(\d+)\]|(\d+))")
_VALU_SDST_FIRST = ("v_readlane_b32", "v_readfirstlane_b32")
_VALU_SDST_SECOND = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32",
                     "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_f32", "v_div_scale_f64")
_VMEM_PREFIX = ("global_", "buffer_", "scratch_", "flat_")


class Inst:
    __slots__ = ("addr", "op", "ops", "text", "ws", "valu", "mfma", "vmem", "sdefs", "vdefs", "vuses", "suses", "sdata", "mwait",
                 "vmcnt", "target", "load_dst", "pk_f32")

    def __init__(self, addr: int, text: str):
        self.addr = addr
        self.text = text
        parts = text.split(None, 1)
        self.op = parts[0]
        self.ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        m = re.match(r"s_nop (\d+)", text)
        self.ws = int(m.group(1)) + 1 if m else 1

    def analyse(self):
        self.valu, self.mfma, self.vmem = _is_valu(self), _is_mfma(self), _is_vmem(self)
        self.sdefs = frozenset(_valu_sgpr_defs(self))
        self.vdefs = frozenset(_vgpr_defs(self))
        self.vuses = frozenset(_vgpr_uses(self))
        su = set()
        if self.vmem:
            for o in self.ops:
                su |= set(_regs(o, "s"))
        self.suses = frozenset(su)
        self.sdata = frozenset(_store_data(self))
        self.mwait = _mfma_wait(self) if self.mfma else 0
        m = re.search(r"vmcnt\((\d+)\)", self.text) if self.op == "s_waitcnt" else None
        self.vmcnt = int(m.group(1)) if m else None
        self.target = _branch_target(self)
        self.load_dst = self.vdefs if self.vmem and "_load_" in self.op and "_lds_" not in self.op else frozenset()
        self.pk_f32 = bool(re.match(r"v_pk_(mul|add|fma)_f32", self.op))


def _regs(operand: str, kind: str) -> List[int]:
    out: List[int] = []
    for m in _REG.finditer(operand):
        if m.group(1) != kind:
            continue
        if m.group(4) is not None:
            out.append(int(m.group(4)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _is_valu(i: Inst) -> bool:
    return i.op.startswith("v_") and not i.op.startswith("v_mfma") and not i.op.startswith("v_smfmac")


def _is_mfma(i: Inst) -> bool:
    return i.op.startswith("v_mfma") or i.op.startswith("v_smfmac")


def _is_vmem(i: Inst) -> bool:
    return i.op.startswith(_VMEM_PREFIX)


def _valu_sgpr_defs(i: Inst) -> List[int]:
    if not _is_valu(i) or not i.ops:
        return []
    base = re.sub(r"_e(32|64)$", "", i.op)
    if base in _VALU_SDST_FIRST or (base.startswith("v_cmp") and i.op.endswith("_e64")):
        return _regs(i.ops[0], "s")
    if base in _VALU_SDST_SECOND and len(i.ops) > 1:
        return _regs(i.ops[1], "s")
    return []


def _vgpr_defs(i: Inst) -> List[int]:
    """VGPRs an instruction writes (destination = first operand for VALU / MFMA / DS reads / VMEM loads / accvgpr_read)."""
    if not i.ops:
        return []
    if i.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_accvgpr_write", "v_nop")):
        return []
    if i.op.startswith("v_"):
        d = _regs(i.ops[0], "v")
        if i.op.startswith(("v_swap", "v_permlane")) and len(i.ops) > 1:      # two destinations
            d += _regs(i.ops[1], "v")
        return d
    if i.op.startswith("ds_read") or i.op.startswith("ds_bpermute") or i.op.startswith("ds_permute") or i.op.startswith("ds_swizzle"):
        return _regs(i.ops[0], "v")
    if _is_vmem(i) and "_load_" in i.op and "_lds_" not in i.op:
        return _regs(i.ops[0], "v")
    return []


def _vgpr_uses(i: Inst) -> List[int]:
    if i.op.startswith(("s_", ";")):
        return []
    start = 0
    if i.op.startswith("v_") and not i.op.startswith(("v_cmp", "v_accvgpr_write")):
        start = 1
    elif i.op.startswith("ds_read") or (_is_vmem(i) and "_load_" in i.op and "_lds_" not in i.op):
        start = 1
    out: List[int] = []
    for o in i.ops[start:]:
        out += _regs(o, "v")
    if _is_mfma(i) and len(i.ops) >= 4:       # D = C accumulate form also reads D's registers through C (already in ops[3])
        pass
    return out


def _store_data(i: Inst) -> List[int]:
    """Data VGPRs of a VMEM store wider than 8 bytes per lane."""
    if not _is_vmem(i) or "_store_" not in i.op or not i.op.endswith(("x3", "x4")):
        return []
    # global_store_dwordx4 vaddr, vdata, saddr|off ; buffer_store_dwordx4 vdata, vaddr|off, srsrc, soffset
    k = 1 if i.op.startswith(("global_", "flat_", "scratch_")) else 0
    return _regs(i.ops[k], "v") if len(i.ops) > k else []


def _mfma_wait(i: Inst) -> int:
    m = re.match(r"v_(?:s?mfmac?)_\w+?_(\d+)x(\d+)x(\d+)", i.op)
    passes = 8
    if m:
        mm = int(m.group(1))
        passes = 8 if mm >= 32 else 4
        if "f64" in i.op:
            passes = 16
    return passes + 3


def parse_kernels(disassembly: str) -> Dict[str, List[Inst]]:
    kernels: Dict[str, List[Inst]] = {}
    cur: Optional[List[Inst]] = None
    for line in disassembly.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body, _, comment = line.strip().partition("//")
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not body.strip() or not am:
            continue
        cur.append(Inst(int(am.group(1), 16), body.strip()))
    return kernels


def _branch_target(i: Inst) -> Optional[int]:
    if not (i.op.startswith("s_cbranch") or i.op == "s_branch") or not i.ops:
        return None
    try:
        simm = int(i.ops[0])
    except ValueError:
        return None
    if simm >= 0x8000:
        simm -= 0x10000
    return i.addr + 4 + 4 * simm


class _State:
    """What the recent past holds on ONE path: producers still inside their hazard windows (wait states since issue), and the
    VMEM operations issued since the oldest register load that no `s_waitcnt vmcnt` has retired yet."""
    __slots__ = ("recent", "vm", "pipe")

    def __init__(self, recent=(), vm=(), pipe=0):
        self.recent = recent        # tuple of (wait states since issue, instruction index), youngest last
        self.vm = vm                # tuple of instruction indices in issue order; vm[0] is a load with a VGPR destination
        self.pipe = pipe            # wait states for which the matrix pipe is still taken by the last MFMA issued

    def key(self):
        return (self.recent, self.vm, self.pipe)


WINDOW = 12


def _step(st: _State, k: int, insts: List[Inst], kernel: str, found: set, r5: bool) -> _State:
    i = insts[k]
    touch_v = i.vuses | i.vdefs
    vm = st.vm
    pipe = st.pipe
    age = st.recent
    if i.mfma and pipe > 0:        # an MFMA issues when the matrix pipe is free: everything older has aged by that stall
        age = tuple((ws + pipe, j) for ws, j in age)
        pipe = 0
    if not r5:
        pass
    elif i.vmcnt is not None:                               # R5: vmcnt retires in issue order, all but the N youngest
        n = i.vmcnt
        vm = vm[len(vm) - n:] if n < len(vm) else vm
        if n == 0:
            vm = ()
        while vm and not insts[vm[0]].load_dst:           # (operations older than every outstanding register load do not matter)
            vm = vm[1:]
    elif vm:
        for j in vm:
            ld = insts[j]
            hit = ld.load_dst & (i.vuses if i.load_dst else touch_v)     # (a younger LOAD into the same register returns in order)
            if hit:
                found.add(f"R5: {kernel}: `{i.text}` @{i.addr:#x} touches v{sorted(hit)} while `{ld.text}` @{ld.addr:#x} is outstanding "
                          f"and no s_waitcnt vmcnt on this path has retired it")
                break
    for ws, j in age:
        p = insts[j]
        need = 0
        if i.vmem and p.sdefs and (p.sdefs & i.suses):                        # R1
            rule, need = "R1", 5
        elif p.sdata and (i.valu or i.op.startswith("v_accvgpr")) and (p.sdata & i.vdefs):      # R2
            rule, need = "R2", 2
        elif i.mfma and p.valu and (p.vdefs & i.vuses):                       # R3
            rule, need = "R3", (PK_F32_TO_MFMA if p.pk_f32 else 2)
        elif p.mfma and not i.mfma and p.vdefs and (p.vdefs & touch_v):      # R4
            rule, need = "R4", p.mwait
        if need and ws < need:
            found.add(f"{rule}: {kernel}: `{p.text}` @{p.addr:#x} -> `{i.text}` @{i.addr:#x}: {ws} wait state(s), {need} required")
    recent = tuple((ws + i.ws, j) for ws, j in age if ws + i.ws < WINDOW)
    pipe = i.mwait - 4 if i.mfma else max(0, pipe - i.ws)          # (passes - 1: its own issue slot is one of them)
    if i.valu and (i.sdefs or i.vdefs) or i.mfma or i.sdata:
        recent = recent + ((0, k),)
    if r5 and i.vmem and (vm or i.load_dst):
        vm = vm + (k,)
        if len(vm) > 96:          # (a load that 96 younger operations have not retired: give up on it rather than grow without bound)
            vm = vm[-96:]
            while vm and not insts[vm[0]].load_dst:
                vm = vm[1:]
    return _State(recent, vm, pipe)


def check_kernel(name: str, insts: List[Inst], max_states: int = 2_000_000) -> List[str]:
    """Every path through the kernel, by a worklist over (instruction, state) with states merged by equality: both sides of
    every conditional branch are followed, so a producer at the end of a loop body meets the consumer at its head."""
    for i in insts:
        i.analyse()
    r5 = any(n in name for n in R5_KERNELS)
    index = {i.addr: k for k, i in enumerate(insts)}
    found: set = set()
    seen = set()
    work = [(0, _State())]
    steps = 0
    while work:
        k, st = work.pop()
        while k < len(insts):
            key = (k, st.key())
            if key in seen:
                break
            seen.add(key)
            steps += 1
            if steps > max_states:
                raise RuntimeError(f"isa_check: {name}: state space larger than {max_states}")
            i = insts[k]
            st = _step(st, k, insts, name, found, r5)
            if i.op in ("s_endpgm", "s_setpc_b64"):
                break
            if i.target is not None and i.target in index:
                if i.op == "s_branch":
                    k = index[i.target]
                    continue
                work.append((index[i.target], st))
            k += 1
    return sorted(found)


def device_disassemblies(so_path: str) -> List[str]:
    """Disassembly text of every gfx950 code object bundled in a shared object."""
    out = []
    with tempfile.TemporaryDirectory(prefix="lush_isa_") as tmp:
        local = os.path.join(tmp, "lib.so")
        os.symlink(os.path.abspath(so_path), local)
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            raise RuntimeError("llvm-objdump --offloading failed:\n" + r.stderr)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            d = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", os.path.join(tmp, f)],
                               capture_output=True, text=True)
            if d.returncode != 0:
                raise RuntimeError("llvm-objdump -d failed:\n" + d.stderr)
            out.append(d.stdout)
    return out


def check_shared_object(so_path: str) -> Tuple[int, List[str]]:
    """(kernels checked, findings).  A shared object without device code is an error (the guard must not pass silently)."""
    n, found = 0, []
    for text in device_disassemblies(so_path):
        for name, insts in parse_kernels(text).items():
            if insts:
                n += 1
                found += check_kernel(name, insts)
    if n == 0:
        raise RuntimeError(f"isa_check: no gfx950 kernels found in {so_path}: the audit cannot vouch for this build")
    return n, found


if __name__ == "__main__":
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblush_march.so")
    n, found = check_shared_object(path)
    print(f"{n} kernels checked, {len(found)} finding(s)")
    for f in found:
        print(" ", f)
    sys.exit(1 if found else 0)
