"""Final validation accuracy of the reference run (30000 steps, B=50) over several initialisation / dropout seeds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import get_data  # noqa: E402
from distributed_tensorflow_amd.models import xor_mlp  # noqa: E402

dev = torch.device("cuda")
accs = []
for seed in range(int(os.environ.get("SEEDS", "6"))):
    xt, yt, xv, yv = get_data(30000, seed=1 + seed, device=dev)
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=dev, seed=seed)
    h = m.fit(xt, yt, epochs=50, batch_size=50, validation_data=(xv, yv), verbose=0)
    accs.append(h.history["val_acc"][-1])
    print("seed %d  val_acc %.4f  train acc %.4f loss %.5f" % (seed, accs[-1], h.history["acc"][-1], h.history["loss"][-1]), flush=True)
print("mean %.4f min %.4f max %.4f" % (sum(accs) / len(accs), min(accs), max(accs)))
