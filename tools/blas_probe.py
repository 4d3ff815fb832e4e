"""What does the vendor SGEMM (torch.matmul fp32 -> hipBLASLt/rocBLAS) reach on this chip for GEMMs of the conv layers' size?
A yardstick for the hand-written implicit-GEMM kernel (not used by the product path)."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
d = torch.device('cuda:0')
for (M, N, K) in [(102400, 128, 1152), (25600, 256, 2304), (6400, 512, 4608), (196608, 256, 2304), (98304, 128, 1152),
                  (102400, 128, 128), (2048, 1024, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=d)
    b = torch.randn(N, K, device=d)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        c = a @ b.t()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10
    print('M=%6d N=%5d K=%5d  %.3f ms  %6.1f TFLOP/s' % (M, N, K, t, 2.0 * M * N * K / t / 1e9), flush=True)
