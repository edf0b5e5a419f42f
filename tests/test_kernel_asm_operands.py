"""The walks' four-entry box test is ONE inline-asm block (ct_pipeline.hip: CT_LINE_NODE, CT_CLOSEST_NODE): per entry three
packed fmas that read the entry's planes from scalar registers, then a compare that writes the entry's wave mask to a scalar
register pair.  The masks must be early-clobber outputs ("=&s"): without the '&' the compiler may give an earlier entry's
mask the register pair of a LATER entry's planes -- legal for it, fatal for the test -- and in one build of round 5 it did
(entry 1's mask landed on entry 3's x planes: shadows missing in 0.1 % of the pixels, found by tools/shadow_tree_check.py
on the GPU; the builds before it were right by the luck of the allocation).  This test looks at the code that was actually
built: in every run of four {fma fma fma ... cmp} groups of the gfx950 code object, no mask is written to registers a later
group of the same run still reads."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FMA = re.compile(r"^\s*v_pk_fma_f32 v\[(56:57|58:59|60:61)\], s\[(\d+):(\d+)\]")
CMP = re.compile(r"^\s*v_cmp_ge_f32_e64 s\[(\d+):(\d+)\], v56, v62")


def blocks_of(lines):
    """runs of four consecutive entries: per entry the scalar pairs its fmas read and the pair its compare writes"""
    entries, reads = [], []
    for line in lines:
        m = FMA.match(line)
        if m:
            reads.append((int(m.group(2)), int(m.group(3))))
            continue
        m = CMP.match(line)
        if m and len(reads) == 3:
            entries.append((reads, (int(m.group(1)), int(m.group(2)))))
            reads = []
    assert len(entries) % 4 == 0, len(entries)
    return [entries[i:i + 4] for i in range(0, len(entries), 4)]


def clobbered(blocks):
    bad = []
    for b, block in enumerate(blocks):
        for k in range(4):
            lo, hi = block[k][1]
            for later in range(k + 1, 4):
                for (rlo, rhi) in block[later][0]:
                    if not (hi < rlo or rhi < lo):
                        bad.append((b, k, later, (lo, hi), (rlo, rhi)))
    return bad


def test_the_checker_sees_the_build_that_went_wrong():
    # the allocation of that build, as it stood in the ISA (masks s[8:9], s[18:19], s[16:17], s[20:21])
    text = """
	v_pk_fma_f32 v[56:57], s[36:37], v[44:45], v[42:43] op_sel_hi:[1,0,0]
	v_pk_fma_f32 v[58:59], s[38:39], v[44:45], v[42:43] op_sel:[0,1,1] op_sel_hi:[1,1,1]
	v_pk_fma_f32 v[60:61], s[40:41], v[46:47], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,0,1]
	v_cmp_ge_f32_e64 s[8:9], v56, v62
	v_pk_fma_f32 v[56:57], s[42:43], v[44:45], v[42:43] op_sel_hi:[1,0,0]
	v_pk_fma_f32 v[58:59], s[44:45], v[44:45], v[42:43] op_sel:[0,1,1] op_sel_hi:[1,1,1]
	v_pk_fma_f32 v[60:61], s[46:47], v[46:47], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,0,1]
	v_cmp_ge_f32_e64 s[18:19], v56, v62
	v_pk_fma_f32 v[56:57], s[48:49], v[44:45], v[42:43] op_sel_hi:[1,0,0]
	v_pk_fma_f32 v[58:59], s[50:51], v[44:45], v[42:43] op_sel:[0,1,1] op_sel_hi:[1,1,1]
	v_pk_fma_f32 v[60:61], s[16:17], v[46:47], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,0,1]
	v_cmp_ge_f32_e64 s[16:17], v56, v62
	v_pk_fma_f32 v[56:57], s[18:19], v[44:45], v[42:43] op_sel_hi:[1,0,0]
	v_pk_fma_f32 v[58:59], s[20:21], v[44:45], v[42:43] op_sel:[0,1,1] op_sel_hi:[1,1,1]
	v_pk_fma_f32 v[60:61], s[22:23], v[46:47], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,0,1]
	v_cmp_ge_f32_e64 s[20:21], v56, v62
"""
    bad = clobbered(blocks_of(text.splitlines()))
    assert bad == [(0, 1, 3, (18, 19), (18, 19))]


def test_no_mask_lands_on_planes_still_to_be_read(tmp_path):
    obj = os.path.join(ROOT, "cobbletrace_amd", "csrc", "ct_pipeline.o")
    if not (os.path.exists(OBJDUMP) and os.path.exists(obj)):
        pytest.skip("no llvm-objdump / no object file (make -C cobbletrace_amd/csrc keeps ct_pipeline.o)")
    local = str(tmp_path / "ct_pipeline.o")
    shutil.copy(obj, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    device = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(device) == 1, os.listdir(tmp_path)
    text = subprocess.run([OBJDUMP, "-d", str(tmp_path / device[0])], check=True, stdout=subprocess.PIPE, text=True).stdout
    blocks = blocks_of(text.splitlines())
    # eight octants x (two shadow-tree kernels + their ordered variants) at the least
    assert len(blocks) >= 32, len(blocks)
    assert clobbered(blocks) == []
