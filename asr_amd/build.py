"""Builds the C-ABI shared library asr_amd/libasr_amd.so from the HIP sources for gfx950.
hipcc cross-compiles without a GPU.  Run:  python -m asr_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libasr_amd.so')


# Per-file compiler flags.  gemm.hip: hipcc's SLP vectoriser turns the operand split's residual subtractions and the
# K-tile joins into v_pk_add_f32 / v_pk_fma_f32, which cost more issue time beside MFMAs than the scalar forms they
# replace (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'): without it the split GEMMs run 4-6 % faster with
# bit-identical results (profiles/r03_gemm_no_slp.txt).
FILE_FLAGS = {'gemm.hip': ['-fno-slp-vectorize']}


def file_flags(src):
    return FILE_FLAGS.get(os.path.basename(src), [])


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def build(force=False, verbose=True, stamps=False):
    """stamps=True builds the diagnostic libasr_amd_stamps.so (in-kernel phase stamps; select it with
    ASR_AMD_LIB=...); never used by tests or bench numbers."""
    global LIB
    srcs = sources()
    if stamps:
        return _build_stamps(srcs)
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in deps):
        return LIB
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-value'] + file_flags(s) + \
              ['-c', s, '-o', o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    return LIB


def _build_stamps(srcs):
    lib = os.path.join(HERE, 'libasr_amd_stamps.so')
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-value',
           '-DSCAN_STAMPS', '-shared', '-o', lib] + srcs
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    if '--stamps' in sys.argv:
        print(build(stamps=True))
        sys.exit(0)
    print(build(force='--force' in sys.argv))
