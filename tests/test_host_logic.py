"""CPU: host-side logic of the product package (graph parser, module tree, state_dict compatibility).  Construction
needs no GPU; running a forward does (there is no fallback), which is asserted too."""
import glob
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import CFG_DIR, GOLDEN, REPO, tiny_cfg


def test_state_dict_keys_match_reference_fixture():
    from models.yolo_test import Model
    for kind in ('fourier', 'add'):
        g = np.load(os.path.join(GOLDEN, 'model_%s_train.npz' % kind))
        m = Model(tiny_cfg(kind))
        assert list(m.state_dict().keys()) == list(g['sd_keys'])
        assert sum(p.numel() for p in m.parameters()) == int(g['n_params'])
        shapes_ok = all(tuple(v.shape) == tuple(o.shape) for v, o in zip(m.state_dict().values(), _oracle(kind).state_dict().values()))
        assert shapes_ok


def _oracle(kind):
    from oracle.ref_model import Model as OModel
    return OModel(tiny_cfg(kind), dropout=0.0)


@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_graph_routing_matches_oracle(kind):
    from models.yolo_test import Model
    m, o = Model(tiny_cfg(kind)), _oracle(kind)
    assert m.save == o.save
    assert len(m.model) == len(o.model)
    for a, b in zip(m.model, o.model):
        assert (a.i, a.f, a.type, a.np) == (b.i, b.f, b.type, b.np)
    det = m.model[-1]
    assert (det.nl, det.na, det.nc, det.no) == (3, 3, tiny_cfg(kind)['nc'], tiny_cfg(kind)['nc'] + 5)
    assert torch.equal(det.stride, torch.tensor([8., 16., 32.]))
    assert torch.allclose(det.anchors, o.model[-1].anchors)
    assert all(bn.eps == 1e-3 and bn.momentum == 0.03 for bn in m.modules() if isinstance(bn, torch.nn.BatchNorm2d))


def test_oracle_state_dict_loads_and_weights_are_ohwi():
    from models.yolo_test import Model
    from oracle import portable_init
    m, o = Model(tiny_cfg('fourier')), _oracle('fourier')
    sd = portable_init.fill_(o.state_dict())
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    w = m.model[1].conv.weight
    assert w.shape[2:] == (3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(w, sd['model.1.conv.weight'])
    # optimiser grouping of train.py:572-579 sees the same module types
    n_bn = sum(isinstance(v, torch.nn.BatchNorm2d) for v in m.modules())
    assert n_bn == sum(isinstance(v, torch.nn.BatchNorm2d) for v in o.modules())


def test_every_shipped_yaml_parses():
    from models.yolo_test import Model
    files = sorted(glob.glob(os.path.join(CFG_DIR, '*.yaml')))
    assert len(files) == 18
    for f in files:
        d = yaml.safe_load(open(f))
        d['depth_multiple'], gw = 0.33, d['width_multiple']
        for row in d['backbone']:
            if row[2] == 'GPT1_fourier':                      # reference quirk B3: the FFM width is not scaled
                row[3] = [int(128 * gw)]
        if os.path.basename(f) == 'yolov5l_fusion_transformer_FLIR_aligned.yaml':
            # the reference file itself carries a typo row `[[14,k], 1, Add, [1]]` (its line 73): unparsable there too
            with pytest.raises(TypeError):
                Model(d)
            continue
        m = Model(d)
        assert m.model[-1].nl == 3


def test_default_yolov5l_parameter_count():
    from models.yolo_test import Model
    m = Model(os.path.join(CFG_DIR, 'yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml'))
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 207.89e6) < 0.01e6, n
    assert len(m.model) == 50 and len(m.state_dict()) == 1593


def test_forward_without_gpu_fails_loudly():
    from models.yolo_test import Model
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    m = Model(tiny_cfg('add'))
    with pytest.raises(Exception):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))


def test_highpass_mask_matches_reference_slicing():
    from mmidet_hip.fusion_ops import highpass_keep_mask
    mask = highpass_keep_mask()
    assert mask == (1 << 64) - 1 - (1 << (3 * 8 + 3))       # everything but unshifted bin (3,3)  


def test_pack_qkv_keeps_parameters_and_state_dict():
    """fusion_ops.pack_qkv re-seats que/key/val_proj on one buffer: same Parameter objects, keys and values; survives
    load_state_dict; a deep copy (ModelEMA) or .to() simply ends up unpacked again."""
    import copy

    import models.common as mc
    from mmidet_hip import fusion_ops as F2
    torch.manual_seed(3)
    gpt = mc.GPT(64, h=8, n_layer=2)
    before = {k: v.clone() for k, v in gpt.state_dict().items()}
    ids = [id(p) for p in gpt.parameters()]
    assert F2.pack_qkv(gpt) == 2
    assert ids == [id(p) for p in gpt.parameters()]
    after = gpt.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
    sa = gpt.trans_blocks[0].sa
    assert F2._back_to_back(sa.que_proj.weight, sa.key_proj.weight, sa.val_proj.weight)
    assert F2._back_to_back(sa.que_proj.bias, sa.key_proj.bias, sa.val_proj.bias)
    assert F2.pack_qkv(gpt) == 2                                           # idempotent
    other = mc.GPT(64, h=8, n_layer=2)
    gpt.load_state_dict(other.state_dict())                                # in-place copies keep the packing
    assert F2._back_to_back(sa.que_proj.weight, sa.key_proj.weight, sa.val_proj.weight)
    assert torch.equal(sa.key_proj.weight, other.trans_blocks[0].sa.key_proj.weight)
    clone = copy.deepcopy(gpt)
    assert torch.equal(clone.trans_blocks[0].sa.val_proj.weight, sa.val_proj.weight)
    sa.key_proj.weight.data.add_(1.0)                                      # a write through one view touches only that third
    assert not torch.equal(sa.key_proj.weight, other.trans_blocks[0].sa.key_proj.weight)
    assert torch.equal(sa.que_proj.weight, other.trans_blocks[0].sa.que_proj.weight)
    assert torch.equal(sa.val_proj.weight, other.trans_blocks[0].sa.val_proj.weight)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts torch.distributed.run itself (and never touches
    the GPU).  Without a GPU here both ranks stop at bench.py's 'needs an MI355X' assertion -- which is the evidence that two
    ranks were started with RANK/WORLD_SIZE set and the parent relayed their failure."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        return
    assert r.returncode != 0
    assert r.stderr.count('bench.py needs an MI355X') >= 2 or 'invalid device ordinal' in r.stderr, r.stderr[-3000:]
    assert '--gpus 2 but WORLD_SIZE' not in r.stderr


def test_fan_out_plan_of_the_default_graph():
    """yolo_test._plan_lanes: every saved map of the yolov5l two-stream-fourier graph that has exactly two consumers is handed
    on by its first consumer (Conv, the fusion transformers' token pooling, nn.Upsample), so no gradient fan-out is left to the
    autograd engine: 16 maps, first consumers of the three kinds, and the plan is dropped by fuse() and survives pickling."""
    import pickle
    import sys
    sys.path[:0] = [REPO, os.path.join(REPO, 'mmi-det_amd')]
    import bench
    from models.common import GPT, Conv
    from models.yolo_test import Model, Upsample2x
    m = Model(bench.load_cfg('l_fourier'))
    plan = m._fan_skip
    assert sum(len(v) for v in plan.values()) == 16
    kinds = {type(m.model[i]).__name__ for i in plan}
    assert kinds == {'Conv', 'GPT', 'GPT1_fourier', 'Upsample2x'}, kinds
    for first, srcs in plan.items():
        layer = m.model[first]
        assert isinstance(layer, (Conv, GPT, Upsample2x)) and getattr(layer, 'fan_skip', False)
        for j in srcs:
            consumers = [l.i for l in m.model if j in m._srcs[l.i]]
            assert len(consumers) == 2 and consumers[0] == first, (j, consumers)
    m2 = pickle.loads(pickle.dumps(m))
    assert m2._fan_skip == plan
    old = dict(m2.__dict__)
    old.pop('_fan_skip')                 # an object pickled by an older build: the plan is rebuilt on load
    m3 = Model.__new__(Model)
    m3.__setstate__(old)
    assert m3._fan_skip == plan


def test_twin_and_concat_plans_of_the_shipped_graphs():
    """yolo_test._plan_twins / _plan_concats (host logic of round 3's launch structure).  Twin plan: every IR-backbone layer is paired
    with the RGB-backbone layer of the same type and parameter shapes whose inputs are themselves a pair (or the two images), pairs
    are one-to-one, and a leader precedes its follower.  Concat plan: a planned Concat's producers are distinct single-use layers
    that can write a channel slice, their offsets tile the buffer exactly, and no planned producer is a twin layer."""
    import sys
    sys.path[:0] = [REPO, os.path.join(REPO, 'mmi-det_amd')]
    import bench
    from models.common import Add, Add2, Concat, Conv, Focus
    from models.yolo_test import Model, Upsample2x
    for name, npairs in (('l_fourier', 14), ('s_add', None)):
        m = Model(bench.load_cfg(name))
        lead, fol = m._leader_of, m._follower_of
        assert len(lead) == len(fol) > 0 and (npairs is None or len(lead) == npairs), (name, len(lead))
        assert {v: k for k, v in fol.items()} == lead                              # one-to-one
        for j, i in lead.items():
            mi, mj = m.model[i], m.model[j]
            assert i < j and type(mi) is type(mj) and not m._lanes[i] and m._lanes[j]
            assert [tuple(p.shape) for p in mi.parameters()] == [tuple(p.shape) for p in mj.parameters()]
            if isinstance(mj, Focus):
                continue
            si, sj = m._srcs[i], m._srcs[j]
            if isinstance(mj, Add2):
                assert lead.get(sj[0]) == si[0] and si[1] == sj[1]
            else:
                assert len(si) == len(sj) == 1 and lead.get(sj[0]) == si[0]
        if name == 's_add':
            assert all(j - i == 10 for j, i in lead.items())                     # fusion_add graphs: row i <-> row i + 10
        # concat plan
        assert m._cat_total, name
        for cat, total in m._cat_total.items():
            assert isinstance(m.model[cat], Concat)
            prods = [(off, j) for j, (c, off) in m._cat_plan.items() if c == cat]
            assert sorted(j for _, j in prods) == sorted(m._srcs[cat])
            width = 0
            for off, j in sorted(prods):
                assert off == width and isinstance(m.model[j], (Conv, Add, Upsample2x)) and j not in lead and j not in fol
                width += _width_of(m, j)
            assert width == total
        m.fuse()
        assert not m._cat_plan and not m.twin                                        # inference graph: plain layers


def _width_of(m, j):
    """channels of a pass-through layer's output (Add / Upsample2x): those of the Conv / C3 / SPP that feeds it."""
    from models.common import C3, SPP, Conv
    while True:
        layer = m.model[j]
        if isinstance(layer, Conv):
            return layer.conv.weight.shape[0]
        if isinstance(layer, C3):
            return layer.cv3.conv.weight.shape[0]
        if isinstance(layer, SPP):
            return layer.cv2.conv.weight.shape[0]
        j = m._srcs[j][0]


def test_rank_core_plan_follows_the_gpus_numa_nodes(tmp_path):
    """bench.py pins each rank before the GPU is touched: ranks share the cores of their GPU's NUMA node (sysfs), equal contiguous
    blocks; without NUMA information equal blocks of the allowed cores in rank order.  A fake sysfs tree of a two-socket, 8-GPU box."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    pci = tmp_path / 'bus' / 'pci' / 'devices'
    for i in range(8):                                   # GPUs 0-3 on node 0, 4-7 on node 1; an unrelated device in between
        d = pci / ('0000:%02x:00.0' % (0x10 + 0x10 * i))
        d.mkdir(parents=True)
        (d / 'vendor').write_text('0x1002\n')
        (d / 'class').write_text('0x120000\n')
        (d / 'numa_node').write_text('%d\n' % (i // 4))
    nic = pci / '0000:05:00.0'
    nic.mkdir()
    (nic / 'vendor').write_text('0x15b3\n')
    (nic / 'class').write_text('0x020000\n')
    (nic / 'numa_node').write_text('0\n')
    for n, cl in ((0, '0-63,128-191'), (1, '64-127,192-255')):
        nd = tmp_path / 'devices' / 'system' / 'node' / ('node%d' % n)
        nd.mkdir(parents=True)
        (nd / 'cpulist').write_text(cl + '\n')
    allowed = set(range(256))
    plans = [bench.plan_rank_cores(r, 8, allowed, sysfs=str(tmp_path)) for r in range(8)]
    assert all(len(p) == 32 for p in plans)
    assert sorted(c for p in plans for c in p) == list(range(256))                  # disjoint, everything used
    node0 = set(range(64)) | set(range(128, 192))
    assert all(set(plans[r]) <= node0 for r in range(4)) and all(not (set(plans[r]) & node0) for r in range(4, 8))
    # a restricted affinity mask (a container's cpuset) is respected
    some = set(range(0, 256, 2))
    assert all(set(bench.plan_rank_cores(r, 8, some, sysfs=str(tmp_path))) <= some for r in range(8))
    # no GPU information (this container): equal contiguous blocks of what is allowed
    assert bench.plan_rank_cores(1, 2, set(range(8)), sysfs=str(tmp_path / 'nothing')) == [4, 5, 6, 7]
    assert bench._cpulist('0-3,8,10-11') == [0, 1, 2, 3, 8, 10, 11]


def test_fused_optimizer_is_a_torch_optimizer_with_sgd_compatible_state():
    """train.py:597 (LambdaLR(optimizer)), 888 (optimizer.state_dict() into the checkpoint), 609 (load_state_dict on resume): the
    fused optimizer takes the scheduler, and its state moves to and from torch.optim.SGD(nesterov).  Host side only (no step)."""
    from mmidet_hip.optim import FusedSGDEMA
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.BatchNorm2d(5))
    groups = [dict(params=[net[1].weight], lr=0.01, momentum=0.9, weight_decay=0.0),
              dict(params=[net[0].weight], lr=0.02, momentum=0.9, weight_decay=5e-4),
              dict(params=[net[0].bias, net[1].bias], lr=0.1, momentum=0.9, weight_decay=0.0)]
    opt = FusedSGDEMA(net, groups)
    assert isinstance(opt, torch.optim.Optimizer) and opt.param_groups[1] is groups[1]
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: 0.5)
    assert [g['lr'] for g in opt.param_groups] == [0.005, 0.01, 0.05] and sch is not None
    ref = torch.optim.SGD([net[1].weight], lr=0.01, momentum=0.9, nesterov=True)
    ref.add_param_group({'params': [net[0].weight], 'weight_decay': 5e-4})
    ref.add_param_group({'params': [net[0].bias, net[1].bias]})
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    ref.step()
    ptr = {p: b.data_ptr() for p, b in opt._bufs.items()}
    opt.load_state_dict(ref.state_dict())
    assert opt._steps == 1                                   # momenta loaded: the first-step form of the kernel is over
    for p, b in opt._bufs.items():
        assert b.data_ptr() == ptr[p] and opt.state[p]['momentum_buffer'] is b        # the kernel's addresses did not move
        assert torch.equal(b, ref.state[p]['momentum_buffer'])
    assert [g['weight_decay'] for g in opt.param_groups] == [0, 5e-4, 0]
    ref2 = torch.optim.SGD([net[1].weight], lr=0.01, momentum=0.9, nesterov=True)
    ref2.add_param_group({'params': [net[0].weight], 'weight_decay': 5e-4})
    ref2.add_param_group({'params': [net[0].bias, net[1].bias]})
    ref2.load_state_dict(opt.state_dict())                   # and back
    for p in net.parameters():
        assert torch.equal(ref2.state[p]['momentum_buffer'], ref.state[p]['momentum_buffer'])


def test_bn_source_hands_its_sums_only_to_the_untouched_gradient_tensor():
    """ops.BnSrc (the host side of mmi_conv_dgrad_bnred): the producer may take the partial sums a consumer's dgrad epilogue left only if
    the gradient it receives IS the tensor that consumer wrote -- same storage address, same version counter, dense.  A second reader
    of the layer's output makes autograd hand over either another tensor (a sum) or the same one accumulated into in place (version
    bumped): both must fall back to the full reduction pass.  The sums are consumed either way."""
    from mmidet_hip import ops
    src = ops.BnSrc(y=None, mi=None, gammas=(None,), betas=(None,), act=1, twin=False)
    dx = torch.zeros(2, 4, 4, 8)
    parts = (torch.zeros(3, 2, 8), 3)
    before = dict(ops.BNRED_COUNT)
    src.wrote(dx, parts)
    assert src.take(dx) is parts and src.parts is None                      # the very tensor
    assert src.take(dx) is None                                             # consumed: nothing left for a second taker
    src.wrote(dx, parts)
    assert src.take(dx.clone()) is None and src.parts is None               # another tensor (autograd summed two gradients)
    src.wrote(dx, parts)
    dx.add_(1.0)                                                            # autograd accumulated a second gradient in place
    assert src.take(dx) is None
    src.wrote(dx, parts)
    assert src.take(dx.permute(0, 3, 1, 2)) is None                         # a strided view of it
    assert ops.BNRED_COUNT['taken'] == before['taken'] + 1 and ops.BNRED_COUNT['rejected'] == before['rejected'] + 3
    # a consumer only counts itself when the feature is on, the layouts agree and the storage is fp32
    x = torch.zeros(1, 2, 2, 8)
    x._bnsrc = ops.BnSrc(None, None, (None,), (None,), 1, twin=False)
    on = ops.BNRED
    try:
        ops.BNRED = False
        assert ops.bn_src_of(x, False) is None and x._bnsrc.uses == 0
        ops.BNRED = True
        assert ops.bn_src_of(x, True) is None and x._bnsrc.uses == 0          # twin consumer of a lane producer
        assert ops.bn_src_of(x, False) is x._bnsrc and x._bnsrc.uses == 1
        assert ops.bn_src_of(torch.zeros(1), False) is None                   # no source hung on the tensor
    finally:
        ops.BNRED = on


def test_round4_planning_entry_points_need_no_gpu():
    """Host-side planners of the round-4 entry points (no device call): grid of the CEM's conv2 kernel (<= 1024 workgroups, equal shares
    of the 32x16 tiles to within one), workspace of its fused weight gradient, row blocks of a dgrad with the BatchNorm reduction."""
    from mmidet_hip import lib
    from mmidet_hip.ops import ConvDesc
    for n, h, w in ((16, 640, 640), (1, 64, 64), (2, 33, 17), (8, 1280, 1280)):
        tiles = n * ((h + 15) // 16) * ((w + 31) // 32)
        g = lib.cem_conv2_fwd_blocks(n, h, w)
        assert 1 <= g <= min(tiles, 1024)
        per = -(-tiles // g)
        assert g * per >= tiles and g * (per - 1) < tiles                 # every workgroup walks `per` or `per - 1` tiles
        tiles16 = n * ((h + 15) // 16) * ((w + 15) // 16)
        assert lib.cem_conv2_wgrad_bn_workspace(n, h, w) == min(tiles16, 2048) * 24 * 27 * 4
    d = ConvDesc(16, 80, 80, 128, 80, 80, 128, 3, 3, 1, 1, 128, 128)
    nb1, nb2 = lib.conv_dgrad_row_blocks_n(d, 1), lib.conv_dgrad_row_blocks_n(d, 2)
    assert nb1 >= 16 * 80 * 80 // 128 and nb2 >= 16 * 80 * 80 // 128        # at least M / 128 row blocks (128-row tiles at most)
    assert nb1 <= 16 * 80 * 80 // 32 and nb2 <= 16 * 80 * 80 // 32
