"""Loss / accuracy per launch of the one-launch step kernel with 1, 2, 4, 8 asynchronous workers (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.models import xor_mlp
dev = torch.device("cuda", 0)
xt, yt, xv, yv = get_data(30000, seed=3, device=dev)
for workers in (1, 2, 4, 8):
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=dev, seed=1)
    plan = m.fused_step_plan(50)
    out = []
    for i in range(6):
        m.stats.zero_()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        m.train_steps_fused(xt, yt, 50, 600, plan, workers=workers)
        t1.record(); torch.cuda.synchronize()
        logs = m.read_stats()
        out.append("%.4f/%.3f/%.0fus" % (logs["loss"], logs.get("binary_rounded_accuracy", logs.get("acc", 0)), t0.elapsed_time(t1) * 1000 / 600))
    print(workers, out, "val", m.evaluate(xv, yv, batch_size=1000), "max|w|", float(m.store.flat.abs().max()),
          "nan", bool(torch.isnan(m.store.flat).any()), flush=True)
