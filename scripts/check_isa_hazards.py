#!/usr/bin/env python3
"""Scan the device code of uformer_amd/lib/libuformer_hip.so for the packed-f32 operand-select forms that measured WRONG on MI355X when an MFMA wave shares
the SIMD (scripts/ubench_hip/pk_opsel.hip, profiles/r04_run19.txt): `v_pk_fma_f32` / `v_pk_add_f32` / `v_pk_mul_f32` whose op_sel makes the LOW result read
the HIGH half of src1 -- op_sel:[x,1,...] -- returned a wrong low result in lanes 48-63 about once in 1e7.  The forms that select on src0 only, and
the op_sel_hi-only broadcast forms, measured clean.  hipcc folds broadcasts and horizontal adds into exactly these forms (the LDS-staged stem lost whole
images to it, profiles/r04_run17.txt / r04_run18.txt), so the build is checked:

    python scripts/check_isa_hazards.py [path/to/lib.so]     -> lists kernel + instruction, exit status 1 if any

The .so embeds one clang offload bundle per translation unit (section .hip_fatbin); the gfx950 code objects are cut out of it here and disassembled with
llvm-objdump."""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
# low-result select of src1 set (op_sel:[x,1...]).  A src2-ONLY select (op_sel:[0,0,1]) is not matched on purpose: scripts/ubench_hip/pk_opsel.hip measured the
# src0 and src2 selects, every op_sel_hi-only form and v_pk_mov_b32 clean (0 mismatches in 1.07e9 results beside MFMA waves, three boxes: profiles/r04_run19.txt)
HAZARD = re.compile(r"\b(v_pk_(?:fma|add|mul)_f32)\b.*\bop_sel:\[[01],1")


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                yield triple, data[pos + off:pos + off + size]
        pos += len(MAGIC)


def scan(path):
    """-> (number of code objects, number of packed-f32 instructions seen, [(kernel, instruction text), ...])"""
    found, n_obj, n_pk = [], 0, 0
    for triple, blob in code_objects(path):
        n_obj += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kernel = "?"
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
                continue
            if "v_pk_" in line:
                n_pk += 1
                if HAZARD.search(line):
                    found.append((kernel, line.strip().split("//")[0].strip()))
    return n_obj, n_pk, found


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uformer_amd", "lib", "libuformer_hip.so")
    n_obj, n_pk, found = scan(path)
    print(f"{path}: {n_obj} gfx950 code objects, {n_pk} packed instructions, {len(found)} with the hazardous operand select")
    for k, ins in found[:40]:
        print(f"  {k[:100]}: {ins}")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
