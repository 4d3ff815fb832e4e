"""Qualification of the opt-in bf16 storage mode without a dataset: train the SAME model
(same initialisation, same stream of synthetic batches, same dropout seeds) for N steps in the fp32 parity mode and in the bf16
storage mode, and set the two loss curves side by side.

    python tools/track_storage_modes.py [--workload l_fourier] [--steps 300] [--batches 24] [--out FILE]

The batch stream cycles over `--batches` distinct synthetic batches (bench.synth with different seeds), so the loss falls as
the model fits them and the curves have a shape to compare; warm-up schedule and hyper-parameters are TrainStep's defaults
(the reference's hyp.scratch values).  Reported: the loss of both runs every 10 steps, the mean over the last 50 steps, the
largest relative gap between the two curves after smoothing over windows of `--batches` steps (one pass over the stream), and
whether anything went non-finite."""
import argparse
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'mmi-det_amd')]
import bench  # noqa: E402
from mmidet_hip import fusion_ops, lib  # noqa: E402
from mmidet_hip.train_step import TrainStep  # noqa: E402
from models.yolo_test import Model  # noqa: E402


def run(storage, args, dev):
    torch.manual_seed(2)
    fusion_ops._seed_state.pop(dev, None)      # the device seed word restarts from torch's seed: both runs draw the same masks
    fusion_ops._drop_counter[0] = 0
    cfg = bench.load_cfg(args.workload)
    size = args.size or bench.IMAGE_SIZE.get(args.workload, 640)
    bs = args.batch or bench.WORKLOADS[args.workload][5]
    model = Model(cfg).to(dev)
    model.storage = storage
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = args.dropout
    model.train()
    lib.set_gemm_precision(5 if storage == 'bf16' else 0)
    ts = TrainStep(model, cfg['nc'], size, bs, accumulate=1, graph=False)
    stream = [bench.synth(bs, size, cfg['nc'], dev, 500 + i) for i in range(args.batches)]
    losses = []
    for it in range(args.steps):
        imgs, tg = stream[it % args.batches]
        loss, _ = ts.step(imgs, tg)
        losses.append(loss.detach())
        if it % 50 == 49:
            torch.cuda.synchronize()
            print('[%s] step %d  loss/bs %.4f' % (storage, it + 1, float(losses[-1]) / bs), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    out = [float(l) / bs for l in losses]
    del ts, model, stream
    torch.cuda.empty_cache()
    lib.set_gemm_precision(0)
    return out


def smooth(v, w):
    return [sum(v[i:i + w]) / w for i in range(0, len(v) - w + 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='l_fourier')
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--batches', type=int, default=24)
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--size', type=int, default=0)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    a = run('f32', args, dev)
    b = run('bf16', args, dev)
    w = args.batches
    sa, sb = smooth(a, w), smooth(b, w)
    gaps = [abs(x - y) / max(abs(x), 1e-12) for x, y in zip(sa, sb)]
    finite = all(x == x and abs(x) != float('inf') for x in a + b)
    tail = min(50, args.steps)
    rep = {
        'workload': args.workload, 'steps': args.steps, 'distinct_batches': args.batches, 'dropout': args.dropout,
        'finite': finite,
        'loss_first': {'f32': a[0], 'bf16': b[0]},
        'loss_mean_last_%d' % tail: {'f32': sum(a[-tail:]) / tail, 'bf16': sum(b[-tail:]) / tail},
        'smoothed_window': w,
        'max_rel_gap_smoothed': max(gaps), 'mean_rel_gap_smoothed': sum(gaps) / len(gaps),
        'max_rel_gap_smoothed_at_step': gaps.index(max(gaps)),
        'every_10_steps': [{'step': i, 'f32': round(a[i], 5), 'bf16': round(b[i], 5)} for i in range(0, args.steps, 10)],
    }
    txt = json.dumps(rep, indent=1)
    print(txt)
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(txt + '\n')


if __name__ == '__main__':
    main()
