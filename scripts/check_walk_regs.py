#!/usr/bin/env python3
"""Checks on the ISA that the register allocator stays out of the walk's hand-managed registers (scripts/gen_walk_bodies.py):
in the kernels that run walk_window6, no compiler-generated instruction -- anything outside an inline-asm block -- may name
v128..v255 or a128..a255.  (Clobber lists do not reserve a register between asm statements; the amdgpu_num_vgpr budget does, and this
is the check that it does what the generator assumes.)

    python scripts/check_walk_regs.py [update.hip]      exit status 0 = clean"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("slice_kernel", "slice_solo_kernel", "scan_kernelILi256E")
LIMIT = 128


def device_asm(src):
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "update.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def violations(text):
    bad = []
    reg = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
    cur, in_asm = None, False
    for ln, line in enumerate(text.splitlines(), 1):
        m = re.match(r"^(_ZN2dq\w+):", line)
        if m:
            cur = m.group(1) if any(k in m.group(1) for k in KERNELS) else None
        if line.startswith(".Lfunc_end"):
            cur = None
        if "#ASMSTART" in line:
            in_asm = True
        if "#ASMEND" in line:
            in_asm = False; continue
        if cur is None or in_asm:
            continue
        code = line.split(";")[0]
        for r in reg.finditer(code):
            hi = int(r.group(2)) if r.group(2) is not None else int(r.group(5))
            if hi >= LIMIT:
                bad.append((cur, ln, line.strip()))
    return bad


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "dqmc_amd", "csrc", "update.hip")
    bad = violations(device_asm(src))
    for k, ln, line in bad[:20]:
        print("%s: line %d: %s" % (k, ln, line))
    print("check_walk_regs: %d compiler-generated uses of hand-managed registers" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
