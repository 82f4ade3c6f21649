#!/usr/bin/env python3
"""Checks the generated code of K1's row-run prefetch (frave_amd/csrc/k1_forward.hip, pf_issue).

The prefetch is an inline-asm `global_load_ubyte` whose completion the compiler does not track (a load it tracked would tie the coefficient stores and the
staging loads to waits for data nobody uses). Its destination register must therefore never be read or written by any other instruction while a load may
be in flight - which is the whole kernel. The kernel keeps ONE register for it (`pf_keep`, read-write in every issue, named by a marker comment at the
kernel's end); this script scans every instantiation of fwd_transform_quant_kernel and fails if that register appears anywhere else than
  * as the destination of the kernel's prefetch loads,
  * in a `v_mov_b32 vN, 0` (its initialisation, on paths that issue nothing before), or
  * in the marker.
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


def scan(asm):
    """(kernels scanned, prefetch loads seen, problems)"""
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
        keep = regs_of(marks[0].split("pf_keep", 1)[1])
        if len(keep) != 1:
            problems.append(f"{name}: marker names {sorted(keep)}")
            return
        (k,) = keep
        kernels += 1
        n = 0
        for no, line in body:
            code = line.split(";", 1)[0].strip() if "pf_keep" not in line else ""
            if not code or code.endswith(":") or code.startswith("."):
                continue
            if k not in regs_of(code):
                continue
            if re.match(rf"global_load_ubyte v{k}, v\d+, s\[\d+:\d+\]$", code):
                n += 1
            elif re.match(rf"v_mov_b32(_e32)? v{k}, 0$", code):
                pass
            else:
                problems.append(f"{name}: line {no}: `{code}` touches the prefetch register v{k}")
        if n == 0:
            problems.append(f"{name}: no prefetch load found")
        loads += n

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
