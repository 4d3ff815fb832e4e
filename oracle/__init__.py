"""CPU oracle for the MMI-Det two-stream hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mmi-det_amd/`` may import this package: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and there
only as the checker / the timed CPU baseline, never as the thing that is shipped.

It is a plain-PyTorch (CPU, fp32) restatement of the reference's algorithm for the path

    models/yolo_test.py  Model.forward / forward_once / parse_model / Detect
    models/common.py     Conv, Bottleneck, C3, SPP, Focus, Concat, Add, Add2, AdaptiveModule3,
                         EnhanceConv2d, GPT, GPT1_fourier, SelfAttention, myTransformerBlock,
                         extract_frequency2, Seperation_loss
    utils/loss.py        ComputeLoss (+ build_targets), utils/general.py bbox_iou (CIoU)

every function citing the reference file:line it follows.  Parity is PINNED: ``oracle/gen_golden.py``
imports the real reference from /root/reference (in the build container only), loads the same portable
hash-initialised weights (``oracle/portable_init.py``) into both, and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those fixtures on every run.
"""
