"""The 2x2 max-pooling backward kernel on the MNIST CNN's second pooling layer: time and effective bandwidth (it writes the
full-resolution gradient, four times the pooled tensors it reads)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from distributed_tensorflow_amd.ops import functional as F
dev = torch.device("cuda", 0)
for B in (4096, 16384):
    shape = (B, 14, 14, 64)
    dy = torch.randn(B, 7, 7, 64, device=dev)
    arg = torch.randint(0, 4, (B, 7, 7, 64), device=dev, dtype=torch.uint8)
    yp = torch.randn(B, 7, 7, 64, device=dev)
    out = torch.empty(shape, device=dev)
    ref = F.maxpool_backward(dy.cpu(), arg.cpu(), shape, (2, 2), yp.cpu(), None, True) if B == 4096 else None
    for _ in range(3):
        F.maxpool_backward(dy, arg, shape, (2, 2), yp, out, True)
    torch.cuda.synchronize()
    if ref is not None:
        print("max diff vs CPU reference:", (out.cpu() - ref).abs().max().item())
    t0 = time.perf_counter()
    for _ in range(20):
        F.maxpool_backward(dy, arg, shape, (2, 2), yp, out, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    nbytes = out.numel() * 4 + dy.numel() * 8 + arg.numel()
    print("B=%d: %.1f us, %.2f TB/s" % (B, dt * 1e6, nbytes / dt / 1e12))
