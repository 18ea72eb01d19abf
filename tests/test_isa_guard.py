"""ISA-level guard for the kernels that hand data over inside a launch (CPU test: reads the built library).

Round 2 found a run-to-run difference in lstm_scan_bwd_ln2d_kernel: hipcc had sunk
`da4 = rs_a * (dah - m1 - ah4 * m2)` behind the step's third publish as two
    v_pk_fma_f32 v[12:13], v[12:13], v[74:75], v[14:15] op_sel:[0,1,0]
    v_pk_fma_f32 v[8:9],  v[10:11], v[74:75], v[8:9]   op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]
(v[74:75] = (m1, m2), written by v_div_fixup_f32 ~200 instructions earlier and read correctly by 16 scalar v_fma_f32
in between; op_sel makes the LOW lane read the pair's HIGH register v75 = m2), and about once per launch the low
lanes (gates 0 and 2) of one 16-lane pass came out as if m2 were 0.  The instruction form alone is exact in a
micro-benchmark, so the trigger is not established; what is established is that no build without a low-lane op_sel
swizzle on a packed fp32 arithmetic instruction has ever shown a differing bit.  The source pins the value
(asm volatile) so that hipcc emits op_sel_hi forms; this test fails when a compiler or code change brings a
low-lane swizzle back into ANY kernel that uses agent-scope (sc1) hand-offs, instead of waiting for the soak test
(tests/test_zy_reproducible.py) to catch the symptom on a GPU."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'asr_amd', 'libasr_amd.so')
OBJDUMP = '/opt/rocm/llvm/bin/llvm-objdump'

PACKED = re.compile(r'\b(v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32)\b.*\bop_sel:\[([01,]+)\]')


def disassemble_kernels():
    """{kernel symbol: [instruction lines]} of the gfx950 code objects bundled in the library."""
    tmp = tempfile.mkdtemp(prefix='asr_isa_')
    try:
        lib = os.path.join(tmp, 'lib.so')
        shutil.copy(LIB, lib)
        subprocess.check_call([OBJDUMP, '--offloading', lib], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if 'gfx950' not in f:
                continue
            out = subprocess.check_output([OBJDUMP, '-d', os.path.join(tmp, f)]).decode()
            cur = None
            for line in out.splitlines():
                m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
                if m:
                    cur = kernels.setdefault(m.group(1), [])
                elif cur is not None and line.strip():
                    cur.append(line.strip())
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason='needs the built library and llvm-objdump')
def test_no_low_lane_swizzle_on_packed_fp32_in_handoff_kernels():
    kernels = disassemble_kernels()
    scans = [k for k in kernels if 'scan_fwd' in k or 'scan_bwd' in k]
    assert len(scans) >= 10, sorted(kernels)[:20]           # the persistent scans are there under their names
    bad = []
    handoff = 0
    for name, ins in kernels.items():
        if not any(re.search(r'\bsc1\b', i) for i in ins):
            continue                                       # no agent-scope hand-off in this kernel
        handoff += 1
        for i in ins:
            m = PACKED.search(i)
            if m and '1' in m.group(2):
                bad.append((name, i))
    assert handoff >= len(scans)
    assert not bad, 'low-lane op_sel swizzle on packed fp32 arithmetic in a hand-off kernel:\n' + \
        '\n'.join('%s: %s' % b for b in bad[:20])
