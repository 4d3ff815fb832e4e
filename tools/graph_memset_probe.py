"""Does a hipMemsetAsync captured into a SINGLE-STREAM hipGraph stay ordered between its neighbouring kernel nodes?

Round 3: the whole training step captured on one stream (no forked lanes, no wgrad side streams) replayed correctly once and
then produced a NaN SSIM term and a wrong objectness loss -- the two outputs whose kernels zero their accumulators with
hipMemsetAsync (mmi_fusion_stats, mmi_detect_loss).  With forked streams in the capture the same graph was right.  This probe
captures  kernel(x += 1) -> memset(x, 0) -> kernel(acc += x)  chains on one stream and replays them: acc must stay 0.
The library no longer uses memset nodes (common.h::mmi_fill_bytes); this file is the record of why.
    python tools/graph_memset_probe.py > profiles/r03_graph_memset_nodes.txt
"""
import ctypes
import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device('cuda:0')
print('torch', torch.__version__, 'hip', torch.version.hip, torch.cuda.get_device_name(0))
for nbytes in (3160, 24576, 1 << 20):
    for chain in (1, 8, 64):
        n = nbytes // 4
        x = torch.ones(n, device=dev)
        acc = torch.zeros(n, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x.add_(1.0)
            acc.add_(x * 0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s = torch.cuda.current_stream().cuda_stream
            for _ in range(chain):
                x.add_(1.0)                                        # kernel node: x != 0
                rc = hip.hipMemsetAsync(x.data_ptr(), 0, nbytes, s)  # memset node
                assert rc == 0, rc
                acc.add_(x)                                        # kernel node: must see zeros
        bad = []
        for r in range(6):
            g.replay()
            torch.cuda.synchronize()
            bad.append(int((acc != 0).sum()))
        print('bytes %8d chain %3d: elements of acc that are not 0 after replays 1..6: %s' % (nbytes, chain, bad))
