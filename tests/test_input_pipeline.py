"""Host input pipeline (csrc/runtime/input_pipeline.cc): every row exactly once per epoch, x/y stay paired, the order
is a function of (seed, epoch), ragged last batch / drop_last, an abandoned epoch, and fit(stream_from_host=True)."""
import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.input_pipeline import HostPipeline


def _data(n=1000, d=7):
    x = torch.arange(n * d, dtype=torch.float32).reshape(n, d)
    y = torch.arange(n, dtype=torch.int64)
    return x, y


@pytest.mark.parametrize("threads", [1, 3])
def test_each_row_once_and_pairs_stay_together(threads):
    x, y = _data()
    pipe = HostPipeline(x, y, batch_size=64, shuffle=True, seed=5, depth=3, threads=threads)
    orders = []
    for epoch in range(3):
        seen = []
        sizes = []
        for xb, yb in pipe.batches(epoch):
            assert torch.equal(xb[:, 0], yb.float() * 7)        # row i of x starts with 7*i
            seen.append(yb.clone())
            sizes.append(len(yb))
        seen = torch.cat(seen)
        assert sizes == [64] * 15 + [40]
        assert torch.equal(seen.sort().values, y)
        assert not torch.equal(seen, y)
        assert [pipe.order(i) for i in range(5)] == seen[:5].tolist()
        orders.append(seen)
    assert not torch.equal(orders[0], orders[1])
    again = HostPipeline(x, y, batch_size=64, shuffle=True, seed=5, depth=4, threads=2)
    assert torch.equal(torch.cat([yb for _, yb in again.batches(1)]), orders[1])     # (seed, epoch) fixes the order
    other = HostPipeline(x, y, batch_size=64, shuffle=True, seed=6)
    assert not torch.equal(torch.cat([yb for _, yb in other.batches(1)]), orders[1])
    for p in (pipe, again, other):
        p.close()


def test_sequential_order_drop_last_and_abandoned_epoch():
    x, y = _data(100, 3)
    pipe = HostPipeline(x, None, batch_size=32, shuffle=False, drop_last=True, depth=2, threads=2)
    got = [xb for xb, yb in pipe.batches(0)]
    assert len(got) == 3 and torch.equal(torch.cat(got), x[:96])
    it = pipe.batches(1)
    next(it)                       # leave the epoch after one batch
    got = [xb for xb, _ in pipe.batches(2)]
    assert torch.equal(torch.cat(got), x[:96])
    pipe.close()
    with pytest.raises(ValueError):
        HostPipeline(x, y[:50], batch_size=8)


def test_fit_streams_from_host_and_learns():
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 2, (4000, 64), generator=g).float()
    yb = (x[:, :32].long() ^ x[:, 32:].long()).float()
    m = models.xor_mlp(dropout=0.0)
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device="cpu", seed=1)
    h = m.fit(x.numpy(), yb.numpy(), epochs=6, batch_size=50, verbose=0, stream_from_host=True, shuffle_seed=3)
    assert len(h.history["loss"]) == 6
    assert h.history["loss"][-1] < h.history["loss"][0] - 0.01
    assert int(m.store.global_step.item()) == 6 * 80


@pytest.mark.gpu
def test_pinned_ring_feeds_the_gpu():
    x, y = _data(5000, 16)
    pipe = HostPipeline(x, y, batch_size=256, shuffle=True, seed=1, device="cuda", depth=4, threads=2)
    for epoch in range(2):
        tot = torch.zeros((), device="cuda", dtype=torch.float64)
        rows = []
        for xb, yb in pipe.batches(epoch):
            assert xb.is_cuda and yb.is_cuda
            tot += xb.double().sum()
            rows.append(yb)
        assert torch.equal(torch.cat(rows).sort().values.cpu(), y)
        assert float(tot) == float(x.double().sum())
    pipe.close()
    m = models.mnist_mlp()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device="cuda", seed=0)
    xs = torch.rand(2048, 784)
    ys = torch.randint(0, 10, (2048,))
    h = m.fit(xs, ys, epochs=2, batch_size=256, verbose=0, stream_from_host=True)
    assert np.isfinite(h.history["loss"][-1]) and int(m.store.global_step.item()) == 16
