"""The reference run (batch 50, 30 000 global steps, Adam 1e-3, dropout 0.3: /root/reference/example.py:12-19,187)
with k asynchronous workers inside one GPU: final validation accuracy over five seeds and examples per second."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.models import xor_mlp
dev = torch.device("cuda", 0)
rows = []
for workers in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    finals, rates = [], []
    for seed in range(5):
        xt, yt, xv, yv = get_data(30000, seed=1 + seed, device=dev)
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=dev, seed=seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = m.fit(xt, yt, epochs=50, batch_size=50, validation_data=(xv, yv), verbose=0, async_workers=workers)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert m.global_step == 30000
        finals.append(h.history["val_acc"][-1])
        rates.append(30000 * 50 / dt)
    row = {"workers": workers, "val_acc_mean": sum(finals) / 5, "val_acc": finals, "examples_per_s_fit_wall_clock": sum(rates) / 5}
    rows.append(row)
    print(json.dumps(row), flush=True)
