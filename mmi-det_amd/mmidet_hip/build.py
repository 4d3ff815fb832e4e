"""Build csrc/*.hip into csrc/libmmidet_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc')
LIB = os.path.join(CSRC, 'libmmidet_hip.so')
# integer/bit-exact kernels must not be FMA-contracted (utils/loss.py:189-245 parity)
# bn.hip: SiLU / SiLU' of the BatchNorm passes on the hardware exp2 / reciprocal (csrc/common.h::MMI_FAST_SILU; one ulp each, the
# parity bound is 1e-3): 25 fewer VALU instructions per element in kernels that run beside the GEMMs -- step 120.10 -> 119.34 ms
# (profiles/r04_fast_silu_step_ab.txt); MMIDET_FAST_SILU=0 at build time keeps expf and the IEEE division
FLAGS = {'targets.hip': ['-ffp-contract=off'],
         'bn.hip': [] if os.environ.get('MMIDET_FAST_SILU') == '0' else ['-DMMI_FAST_SILU=1']}


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    deps = srcs + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(CSRC, '..', '..', 'include', 'mmidet_hip.h')]
    objs = []
    newest = max(os.path.getmtime(p) for p in deps)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    hdr_m = max(os.path.getmtime(p) for p in deps if not p.endswith('.hip'))
    todo = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), hdr_m):
            continue
        todo.append(['hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-c', s, '-o', o] + FLAGS.get(os.path.basename(s), []))
    # at most MAXJ compilers at once (a hipcc of one of the kernel-template units takes 1-2 GB; the build box has 8 cores / 64 GB);
    # the heaviest units (the implicit-GEMM instantiations) go first so that they overlap with everything else
    todo.sort(key=lambda c: -os.path.getsize(c[c.index('-c') + 1]) - (1 << 30) * ('igemm' in os.path.basename(c[c.index('-c') + 1])))
    maxj = int(os.environ.get('MMIDET_BUILD_JOBS', '12'))
    running = []
    while todo or running:
        while todo and len(running) < maxj:
            cmd = todo.pop(0)
            if verbose:
                print(' '.join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        for item in list(running):
            rc = item[1].poll()
            if rc is None:
                continue
            running.remove(item)
            if rc != 0:
                for _, p in running:
                    p.kill()
                raise RuntimeError('hipcc failed: ' + ' '.join(item[0]))
        if running:
            import time
            time.sleep(0.2)
    cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
