#!/usr/bin/env python3
"""Checks the generated code of K1's row-run prefetch (frave_amd/csrc/k1_forward.hip, pf_issue).

The prefetch is an inline-asm `global_load_ubyte` whose completion the compiler does not track (a load it tracked would tie the coefficient stores and the
staging loads to waits for data nobody uses). Its destination register must therefore never be read or written by any other instruction while a load may
be in flight - which is the whole kernel. The kernel keeps ONE register for it (`pf_keep`, read-write in every issue, named by a marker comment at the
kernel's end); this script scans every instantiation of fwd_transform_quant_kernel and fails if that register appears anywhere else than
  * the prefetch loads themselves and the register's initialisation (`v_mov_b32 vN, 0`, inline asm too)
are the only instructions that WRITE it (a compiler-made read can only be a don't-care operand, e.g. the unused high half of a 64-bit addend: the
program never uses the loaded byte), and that all prefetch loads of a kernel land in ONE register (a split live range would free one of them early).
The Makefile runs it on the assembly made with the object's exact flags before it links libfri_hip.so; tests/test_k2_isa.py runs it in the CPU suite.

    python tools/check_k1_isa.py                 # make the assembly (csrc/build/k1_forward.s), then scan it; exit code 0 = clean
    python tools/check_k1_isa.py <file.s>        # scan an existing assembly file
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "frave_amd", "csrc")


def make_assembly():
    asm = os.path.join(CSRC, "build", "k1_forward.s")
    subprocess.run(["make", "-s", "-C", CSRC, asm], check=True)
    return asm


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


# registers an instruction WRITES: the first operand of vector ALU / load instructions (stores and compares into SGPRs write no VGPR), both operands of the swaps
def written(code):
    m = re.match(r"(\S+)\s+(.*)$", code)
    if not m:
        return set()
    op, rest = m.group(1), m.group(2)
    if op.startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "s_", "v_cmp", "v_readfirstlane", "v_readlane", "global_atomic", "buffer_atomic")) and "_rtn" not in op and " glc" not in rest:
        return set()
    if not op.startswith(("v_", "ds_", "global_", "buffer_", "scratch_", "flat_")):
        return set()
    ops = [o.strip() for o in rest.split(",")]
    w = regs_of(ops[0]) if ops else set()
    if op.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")) and len(ops) > 1:
        w |= regs_of(ops[1])
    return w


def scan(asm):
    """(kernels scanned, prefetch loads seen, problems). Per kernel: K = the destination registers of the inline-asm prefetch loads; nothing else may WRITE a
    register of K (a read can only be a don't-care operand - the high half of a 64-bit addend, say: the program never uses the loaded byte)."""
    kernels, loads, problems = 0, 0, []
    name, body = None, []

    def finish():
        nonlocal kernels, loads
        if name is None:
            return
        marks = [l for _, l in body if "; pf_keep" in l]
        if len(marks) != 1:
            problems.append(f"{name}: {len(marks)} pf_keep markers (expected 1)")
            return
        keep = set(regs_of(marks[0].split("pf_keep", 1)[1]))
        in_asm, n = False, 0
        for _, line in body:  # pass 1: every register an asm-block prefetch load lands in
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif in_asm:
                m = re.match(r"\s*global_load_ubyte (v\d+), v\d+, s\[\d+:\d+\]\s*$", line.split(";", 1)[0])
                if m:
                    keep |= regs_of(m.group(1))
                    n += 1
        if n == 0:
            problems.append(f"{name}: no prefetch load found")
            return
        if len(keep) != 1:
            problems.append(f"{name}: the prefetch loads land in {sorted(keep)}: the register was split")
        kernels += 1
        loads += n
        in_asm = False
        for no, line in body:  # pass 2: who else writes them?
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";", 1)[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            hit = written(code) & keep
            if not hit:
                continue
            if in_asm and (re.match(r"global_load_ubyte v\d+, v\d+, s\[\d+:\d+\]$", code) or re.match(r"v_mov_b32(_e32)? v\d+, 0$", code)):
                continue
            problems.append(f"{name}: line {no}: `{code}` writes the prefetch register v{sorted(hit)[0]}")

    with open(asm) as f:
        for no, line in enumerate(f, 1):
            m = re.match(r"^(_ZN3fri\S*fwd_transform_quant_kernel\S*):", line)
            if m:
                finish()
                name, body = m.group(1), []
            elif name is not None:
                body.append((no, line.rstrip("\n")))
                if "s_endpgm" in line:
                    finish()
                    name, body = None, []
    finish()
    return kernels, loads, problems


def main():
    asm = sys.argv[1] if len(sys.argv) > 1 else make_assembly()
    kernels, loads, problems = scan(asm)
    for p in problems:
        print("K1 ISA:", p)
    print(f"{kernels} forward kernel instantiations checked, {loads} prefetch loads, {len(problems)} problem(s)")
    if kernels < 8:
        print("K1 ISA: fewer instantiations than the library launches - the scan no longer matches the kernel")
        return 1
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
