import torch, sys
sys.path.insert(0, '.')
from distributed_tensorflow_amd.ops import functional as F
dev = torch.device('cuda')
for (M, K, N) in [(50, 64, 128), (128, 32, 128), (128, 64, 128), (64, 32, 64), (50, 128, 32), (256, 784, 128)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev)
    w = (torch.rand(K, N, generator=g) * 2 - 1).to(dev)
    y = F.dense_forward(x, w, None, 0)
    ref = x.double() @ w.double()
    err = (y.double() - ref).abs()
    print(M, K, N, 'max err', err.max().item(), 'bad rows', (err.max(1).values > 1e-3).nonzero().flatten()[:10].tolist(),
          'bad cols', (err.max(0).values > 1e-3).nonzero().flatten()[:10].tolist())
    # which k contributions are missing: use one-hot rows
    if M == 128 and K == 32:
        for k in range(0, 32):
            xe = torch.zeros(M, K, device=dev); xe[:, k] = 1
            ye = F.dense_forward(xe, w, None, 0)
            e = (ye - w[k].unsqueeze(0)).abs().max().item()
            if e > 1e-4: print('  k', k, 'err', e)
