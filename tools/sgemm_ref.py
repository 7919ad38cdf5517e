"""Context for the conv roofline: what the vendor GEMM library (hipBLASLt / rocBLAS behind torch.matmul, fp32, TF32 off) reaches on this
chip for plain dense fp32 GEMMs of the shapes the implicit-GEMM convs have (M = output pixels of 64 images, N = Cout, K = taps x Cin)."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
shapes = [("8192^3", 8192, 8192, 8192), ("conv3 x6 (M=64*56*56, N=128, K=1024)", 200704, 128, 1024), ("conv4 (M=64*28*28, N=256, K=12288)", 50176, 256, 12288),
          ("deconv3 phase (M=64*56*56, N=64, K=1024)", 200704, 64, 1024), ("conv2 (M=64*112*112, N=64, K=512)", 802816, 64, 512)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(K, N, device=dev)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        c = a @ b
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:48s} {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s  ({2.0 * M * N * K / ms / 1e9 / 157.3 * 100:.0f} % of 157.3)")
