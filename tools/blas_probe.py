"""What the vendor GEMM libraries reach on the ViT-Base / CSWin GEMM shapes (fp16 in, fp16 out, via torch.matmul -> hipBLASLt / rocBLAS):
a yardstick for csrc/gemm16.hip, not a product path.    python tools/blas_probe.py"""
import torch, time
dev = torch.device("cuda", 0)
shapes = [("vit qkv", 50432, 2304, 768), ("vit proj", 50432, 768, 768), ("vit fc1", 50432, 3072, 768), ("vit fc2", 50432, 768, 3072),
          ("cswin s3 qkv", 50176, 768, 256), ("cswin s3 fc1", 50176, 1024, 256), ("cswin s3 fc2", 50176, 256, 1024), ("mixer fc1", 50176, 2048, 512),
          ("xcit qkv", 50176, 1152, 384), ("xcit proj", 50176, 384, 384), ("xcit fc1", 50176, 1536, 384), ("xcit fc2", 50176, 384, 1536),
          ("cswin s3 proj", 50176, 256, 256), ("mixer tok fc1", 131072, 256, 256), ("mixer fc2", 50176, 512, 2048)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) / K ** 0.5
    for _ in range(3): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): torch.nn.functional.linear(a, w)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("%-14s M=%d N=%d K=%d  %.4f ms  %.0f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
