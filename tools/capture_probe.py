"""Which stream-fork patterns survive hipStreamBeginCapture/EndCapture on this ROCm?  (VERDICT r1 "weak" 5: the round-1
capture_end segfault with the twin lanes AND the per-layer dgrad||wgrad fork in one capture.)

Every pattern runs in its OWN child process (a crash is the datum, not the end of the probe), captures a small graph of
plain torch kernels with the given fork/join topology, replays it and checks the numbers.  The `step:*` patterns capture
the real TrainStep of the tiny FFM graph with the work-around of train_step.py::_capture switched off.

    python tools/capture_probe.py            # all patterns, one line each
    python tools/capture_probe.py nested     # one pattern in this process
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mmi-det_amd'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

PATTERNS = ['flat', 'nested', 'nested_x300', 'nested_lazy_stream', 'nested_join_to_origin', 'nested_two_children',
            'nested_two_children_join_all', 'wait_on_idle_outside_stream', 'step:lanes', 'step:wgrad', 'step:lanes+wgrad_nside1',
            'step:lanes+wgrad', 'step:lanes+wgrad_deferred']


def torch_pattern(name):
    import torch
    dev = torch.device('cuda:0')
    a = torch.randn(256, 256, device=dev)
    outs = []

    def work(x):
        return torch.tanh(x @ a)

    main = torch.cuda.Stream()
    A, B, C = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    idle = torch.cuda.Stream()

    def body():
        x = work(a)
        if name == 'flat':
            A.wait_stream(main)
            B.wait_stream(main)
            with torch.cuda.stream(A):
                ya = work(x)
            with torch.cuda.stream(B):
                yb = work(x)
            main.wait_stream(A)
            main.wait_stream(B)
            outs[:] = [ya, yb]
            return
        reps = 300 if name == 'nested_x300' else 1
        for _ in range(reps):
            A.wait_stream(main)                                   # lane fork
            with torch.cuda.stream(A):
                ya = work(x)
                b = torch.cuda.Stream() if name == 'nested_lazy_stream' else B
                b.wait_stream(A)                                  # nested fork (wgrad stream of the lane)
                with torch.cuda.stream(b):
                    yb = work(ya)
                yc = None
                if name.startswith('nested_two_children'):
                    C.wait_stream(A)
                    with torch.cuda.stream(C):
                        yc = work(ya + 1)
                ya2 = work(ya)
                if name in ('nested_join_to_origin', 'wait_on_idle_outside_stream'):
                    pass                                          # B is joined by the origin below, not by its parent lane
                elif name == 'nested_two_children_join_all':
                    A.wait_stream(b)
                    A.wait_stream(C)
                    A.wait_stream(b)                              # (a redundant second join, as _join_side over NSIDE streams does)
                else:
                    A.wait_stream(b)
                    if yc is not None:
                        A.wait_stream(C)
            main.wait_stream(A)
            if name in ('nested_join_to_origin', 'wait_on_idle_outside_stream'):
                main.wait_stream(b)
            if name == 'wait_on_idle_outside_stream':
                main.wait_stream(idle)                            # a stream that is NOT part of the capture
            x = work(ya2 + yb + (yc if yc is not None else 0))
        outs[:] = [x]

    with torch.cuda.stream(main):
        body()
        torch.cuda.synchronize()
        want = [o.clone() for o in outs]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
        g.replay()
        torch.cuda.synchronize()
        for o, w in zip(outs, want):
            assert torch.allclose(o, w, rtol=1e-4, atol=1e-5), float((o - w).abs().max())
    print('OK')


def step_pattern(name):
    import torch
    from mmidet_hip import ops
    import test_step_gpu as T
    flags = name.split(':')[1]
    os.environ['MMIDET_TWO_STREAMS'] = '1' if 'lanes' in flags else '0'
    ops.OVERLAP_WGRAD = 'wgrad' in flags
    if 'nside1' in flags:
        ops.NSIDE = 1
    m1, ts1, cfg = T.make(graph=True, defer_join='deferred' in flags)
    m2, ts2, _ = T.make(graph=False)
    ts1._capture_keeps_wgrad_overlap = True          # the probe's point: no work-around
    b = T.batch(cfg, 20)
    for _ in range(2):
        ts2.step(*b)
    l1, _ = ts1.step(*b)
    l2, _ = ts2.step(*b)
    torch.cuda.synchronize()
    err = float((l1 - l2).abs().max() / l2.abs().max())
    assert err < 1e-3, err
    l1, _ = ts1.step(*b)
    torch.cuda.synchronize()
    assert torch.isfinite(l1).all()
    print('OK rel %.2e' % err)


def main():
    if len(sys.argv) > 1:
        import faulthandler
        faulthandler.enable()                      # a SIGSEGV prints the Python frame it happened under
        name = sys.argv[1]
        (step_pattern if name.startswith('step:') else torch_pattern)(name)
        return
    for name in PATTERNS:
        env = dict(os.environ)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=300, env=env)
            tail = (r.stdout.strip().splitlines() or [''])[-1]
            err = ''
            if r.returncode != 0:
                lines = [l for l in r.stderr.strip().splitlines() if l.strip()]
                key = [l.strip() for l in lines if 'Error' in l or 'error' in l or 'Fatal' in l or 'HIP' in l or 'File "' in l]
                err = ' | '.join((key or lines)[:3])[:400]
            print('%-34s rc=%-4d %s %s' % (name, r.returncode, tail, err), flush=True)
        except subprocess.TimeoutExpired:
            print('%-34s TIMEOUT' % name, flush=True)


if __name__ == '__main__':
    main()
