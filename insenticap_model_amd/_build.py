"""Builds libinsenticap_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m insenticap_model_amd._build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libinsenticap_hip.so')
SOURCES = ['gemm_f32.hip', 'attention.hip', 'pointwise.hip', 'backward.hip', 'step.hip', 'rows.hip']
HEADERS = [os.path.join(CSRC, 'common.h'),
           os.path.join(os.path.dirname(HERE), 'include', 'insenticap_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


LAST_BUILD = {'compiled': [], 'reused': [], 'linked': False}      # what the last build() call actually did


def build(force=False, verbose=False):
    """mtime-driven: an object is recompiled when its source or a header is newer (or `force`); LAST_BUILD records which
    translation units were compiled and which in-tree objects were reused, so a caller can say which it was."""
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    LAST_BUILD.update(compiled=[], reused=[], linked=False)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIB_DIR, s.replace('.hip', '.o'))
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
            LAST_BUILD['compiled'].append(s)
        else:
            LAST_BUILD['reused'].append(s)
        objs.append(obj)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        LAST_BUILD['linked'] = True
    if build_cider(force, verbose, _report=True):
        LAST_BUILD['compiled'].append('cider.cpp')
    else:
        LAST_BUILD['reused'].append('cider.cpp')
    return LIB_PATH


CIDER_LIB_PATH = os.path.join(LIB_DIR, 'libinsenticap_cider.so')


def build_cider(force=False, verbose=False, _report=False):
    """Host-side CIDEr-D reward library (plain C++, g++)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    src = os.path.join(CSRC, 'cider.cpp')
    hdr = os.path.join(os.path.dirname(HERE), 'include', 'insenticap_cider.h')
    did = False
    if force or _stale(CIDER_LIB_PATH, [src, hdr]):
        cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wall', src, '-o', CIDER_LIB_PATH]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        did = True
    return did if _report else CIDER_LIB_PATH


def resource_report():
    """Per-kernel register / scratch / LDS usage from hipcc's -Rpass-analysis (compile only).
    Returns {kernel: {'vgprs', 'agprs', 'scratch', 'vgpr_spill', 'occupancy'}}."""
    import re
    hipcc = _hipcc()
    rep = {}
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        out = subprocess.run([hipcc] + FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o',
                                                 os.devnull], capture_output=True, text=True, check=True).stderr
        cur = None
        for line in out.splitlines():
            m = re.search(r'remark: (.*?)\s*\[-Rpass', line)
            if not m:
                continue
            txt = m.group(1).strip()
            if txt.startswith('Function Name:'):
                cur = rep.setdefault(txt.split(':', 1)[1].strip(), {})
            elif cur is not None:
                for key, name in (('VGPRs:', 'vgprs'), ('AGPRs:', 'agprs'), ('ScratchSize [bytes/lane]:', 'scratch'),
                                  ('VGPRs Spill:', 'vgpr_spill'), ('Occupancy [waves/SIMD]:', 'occupancy')):
                    if txt.startswith(key):
                        cur[name] = int(txt.split(':')[-1])
    return rep


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
