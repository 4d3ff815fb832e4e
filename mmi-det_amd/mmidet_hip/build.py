"""Build csrc/*.hip into csrc/libmmidet_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc')
LIB = os.path.join(CSRC, 'libmmidet_hip.so')
# integer/bit-exact kernels must not be FMA-contracted (utils/loss.py:189-245 parity)
FLAGS = {'targets.hip': ['-ffp-contract=off']}


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    deps = srcs + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(CSRC, '..', '..', 'include', 'mmidet_hip.h')]
    objs = []
    newest = max(os.path.getmtime(p) for p in deps)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    hdr_m = max(os.path.getmtime(p) for p in deps if not p.endswith('.hip'))
    procs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), hdr_m):
            continue
        cmd = ['hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-c', s, '-o', o] + \
            FLAGS.get(os.path.basename(s), [])
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
