#!/usr/bin/env python
"""Divergence of the one-launch step kernel from the layer-by-layer path, per variable, after n steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import get_data  # noqa: E402
from distributed_tensorflow_amd.models import xor_mlp  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    opt = os.environ.get("OPT", "adam")
    xt, yt = get_data(4000, val_set_size=0, seed=2, device=dev)[:2]
    for n in (1, 2, 5, 10, 20, 40, 80):
        ms = []
        for _ in range(2):
            m = xor_mlp()
            m.compile(loss="mean_squared_error", optimizer=opt, metrics=["binary_rounded_accuracy"], device=dev, seed=5)
            ms.append(m)
        a, b = ms
        for la, lb in zip(a.layers, b.layers):
            if hasattr(la, "seed"):
                lb.seed = la.seed
        for i in range(n):
            a.train_step(xt[i * 50:(i + 1) * 50], yt[i * 50:(i + 1) * 50])
        if os.environ.get("SPLIT") and n > 1:
            b.train_steps_fused(xt, yt, 50, n // 2)
            b.train_steps_fused(xt[(n // 2) * 50:], yt[(n // 2) * 50:], 50, n - n // 2)
        else:
            b.train_steps_fused(xt, yt, 50, n)
        torch.cuda.synchronize()
        line = ["n=%3d" % n]
        for va, vb in zip(a.store.trainable, b.store.trainable):
            d = (va.tensor - vb.tensor).abs()
            line.append("%s %.2e (%d>1e-5)" % (va.name.replace("dense_", "d"), d.max().item(), int((d > 1e-5).sum())))
        for k in a.store.slots:
            d = (a.store.slots[k] - b.store.slots[k]).abs()
            line.append("slot %s %.2e" % (k, d.max().item()))
        print("  ".join(line))


def reuse_case():
    """80 steps where steps 40..79 see the batches of steps 0..39 again; fused in two launches, layer-wise eager, and
    layer-wise with gradients perturbed at the 1e-7 level (how far does rounding noise alone drift?)."""
    dev = torch.device("cuda", 0)
    xt, yt = get_data(2000, val_set_size=0, seed=2, device=dev)[:2]
    ms = []
    for _ in range(3):
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=dev, seed=5)
        ms.append(m)
    a, b, c = ms
    for la, lb, lc in zip(a.layers, b.layers, c.layers):
        if hasattr(la, "seed"):
            lb.seed = lc.seed = la.seed
    torch.manual_seed(0)
    for ep in range(2):
        for i in range(40):
            xb, yb = xt[i * 50:(i + 1) * 50], yt[i * 50:(i + 1) * 50]
            a.train_step(xb, yb)
            c.compute_gradients(xb, yb)
            c.store.flat_grad.mul_(1 + 2e-7 * torch.randn_like(c.store.flat_grad))
            c.strategy.apply_gradients(c)
        b.train_steps_fused(xt, yt, 50, 40)
        torch.cuda.synchronize()
        print("epoch %d: fused-vs-layerwise %.3e   perturbed-layerwise-vs-layerwise %.3e" %
              (ep, (a.store.flat - b.store.flat).abs().max().item(), (a.store.flat - c.store.flat).abs().max().item()))


def fit_case():
    dev = torch.device("cuda", 0)
    xt, yt = get_data(2000, val_set_size=0, seed=2, device=dev)[:2]
    for epochs in (1, 2):
        ms = []
        for _ in range(2):
            m = xor_mlp()
            m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=dev, seed=5)
            ms.append(m)
        a, b = ms
        for la, lb in zip(a.layers, b.layers):
            if hasattr(la, "seed"):
                lb.seed = la.seed
        ms2 = []
        for _ in range(2):
            m = xor_mlp()
            m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=dev, seed=5)
            for la, lb in zip(a.layers, m.layers):
                if hasattr(la, "seed"):
                    lb.seed = la.seed
            ms2.append(m)
        c, dd = ms2
        for ep in range(epochs):
            for i in range(40):
                c.train_step(xt[i * 50:(i + 1) * 50], yt[i * 50:(i + 1) * 50])
            dd.train_steps_fused(xt, yt, 50, 40)
        ha = a.fit(xt, yt, epochs=epochs, batch_size=50, shuffle=False, verbose=0, fused=False)
        hb = b.fit(xt, yt, epochs=epochs, batch_size=50, shuffle=False, verbose=0, fused=True)
        f = lambda u, v: (u.store.flat - v.store.flat).abs().max().item()
        print("  manual-layer vs manual-fused %.2e | fit-layer vs manual-layer %.2e | fit-fused vs manual-fused %.2e" %
              (f(c, dd), f(a, c), f(b, dd)))
        d = (a.store.flat - b.store.flat).abs()
        print("fit epochs=%d: max diff %.3e  steps %d/%d  loss %s / %s" %
              (epochs, d.max().item(), a.global_step, b.global_step, ha.history["loss"], hb.history["loss"]))


if __name__ == "__main__":
    if os.environ.get("REUSE"):
        reuse_case()
    elif os.environ.get("FIT"):
        fit_case()
    else:
        main()
