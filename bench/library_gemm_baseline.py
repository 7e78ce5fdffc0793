"""The two first-layer products of the headline model on the vendor library (torch.mm -> hipBLASLt / rocBLAS, fp32), for
comparison with the hand-written kernels (which also fuse bias, ReLU, dropout / the bias gradient)."""
import torch

dev = torch.device("cuda")
M, K, N = 262144, 784, 128
x = torch.rand(M, K, device=dev) * 2 - 1
w = torch.rand(K, N, device=dev) * 2 - 1
dz = torch.rand(M, N, device=dev) * 2 - 1
torch.backends.cuda.matmul.allow_tf32 = False


def t(fn, name, flops):
    for _ in range(30):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1000 / 20
    print("%-40s %7.1f us  %6.1f TFLOP/s" % (name, us, flops / us / 1e6), flush=True)


y = torch.empty(M, N, device=dev)
dw = torch.empty(K, N, device=dev)
t(lambda: torch.mm(x, w, out=y), "library forward  x[M,784] . w[784,128]", 2.0 * M * K * N)
t(lambda: torch.mm(x.t(), dz, out=dw), "library wgrad    x^T[784,M] . dz[M,128]", 2.0 * M * K * N)
h = torch.rand(M, N, device=dev)
w2 = torch.rand(N, N, device=dev)
y2 = torch.empty(M, N, device=dev)
t(lambda: torch.mm(h, w2, out=y2), "library forward  h[M,128] . w[128,128]", 2.0 * M * N * N)
