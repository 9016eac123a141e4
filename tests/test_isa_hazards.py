"""The built library must not contain the packed-f32 operand-select forms that measured wrong on MI355X beside MFMA waves (scripts/check_isa_hazards.py has
the finding; scripts/ubench_hip/pk_opsel.hip is the reproducer): hipcc produces them by itself when it folds a broadcast or a horizontal add into a
packed instruction, so the DISASSEMBLY of uformer_amd/lib/libuformer_hip.so is what is checked, on the CPU, after every build."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_no_hazardous_packed_operand_selects_in_the_library():
    import check_isa_hazards as chk
    lib = os.environ.get("UFORMER_HIP_LIB", os.path.join(ROOT, "uformer_amd", "lib", "libuformer_hip.so"))
    if not os.path.exists(lib) or not os.path.exists(chk.OBJDUMP):
        pytest.skip("library or llvm-objdump not present")
    n_obj, n_pk, found = chk.scan(lib)
    assert n_obj >= 8 and n_pk > 1000, (n_obj, n_pk)                    # the scan really saw the device code
    assert not found, "hazardous packed-f32 operand selects:\n" + "\n".join(f"{k}: {i}" for k, i in found[:20])


def test_the_scanner_recognises_the_hazardous_forms():
    import check_isa_hazards as chk
    bad = ["v_pk_fma_f32 v[10:11], v[50:51], v[44:45], v[10:11] op_sel:[0,1,0]", "v_pk_add_f32 v[250:251], v[250:251], v[250:251] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,1] op_sel_hi:[1,1,0]"]
    good = ["v_pk_fma_f32 v[12:13], v[48:49], v[44:45], v[12:13] op_sel_hi:[1,0,1]", "v_pk_fma_f32 v[10:11], v[6:7], v[8:9], v[0:1] op_sel:[1,0,0] op_sel_hi:[0,1,1]",
            "v_pk_mov_b32 v[8:9], v[4:5], v[6:7] op_sel:[1,0]", "v_pk_fma_f32 v[8:9], v[6:7], s[4:5], v[2:3]", "v_pk_mul_f32 v[8:9], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[0,1]"]
    assert all(chk.HAZARD.search(b) for b in bad) and not any(chk.HAZARD.search(g) for g in good)
