"""Fault-style test of the staleness bound: one worker is slowed down on purpose.  With ``max_staleness=2`` the fast
worker is held back (its clock never leads the slowest clock by more than the bound when it starts a step); with the
reference's unbounded mode (/root/reference/example.py:168-170: plain optimizer, no coordination) it runs far ahead."""
import json
import subprocess
import sys

import pytest

from .test_example_cluster import ROOT, _env, _ports

WORKER = r'''
import json, sys, time, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.models import xor_mlp
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

index, hosts, steps, bound = int(sys.argv[1]), sys.argv[2].split(","), int(sys.argv[3]), int(sys.argv[4])
srv = Server(ClusterSpec({"worker": hosts}), "worker", index, config={"device": "cpu"})
strat = AsyncPSStrategy(srv, max_staleness=None if bound < 0 else bound)
m = xor_mlp()
m.compile(loss="mean_squared_error", optimizer="adam", strategy=strat, device="cpu", seed=index)
x, y = get_data(50 * steps, val_set_size=0, seed=index)[:2]
x, y = torch.as_tensor(x, dtype=torch.float32), torch.as_tensor(y, dtype=torch.float32)
lead_at_start = []
for i in range(steps):
    if index == 1:
        time.sleep(0.05)                       # the straggler
    m.train_step(x[i * 50:(i + 1) * 50], y[i * 50:(i + 1) * 50])
    # how far this worker's clock is ahead of the slowest one, right after the step it was allowed to take
    lead_at_start.append(strat.clock - 1 - strat.min_clock())
strat.worker_done()
strat.cp0.barrier("all_done", len(hosts), timeout_s=120)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"index": index, "max_lead": max(lead_at_start), "hist": strat.staleness_hist}))
'''


def _run(tmp_path, bound, steps=40):
    n = 2
    ports = _ports(n + 1)
    hosts = ",".join("127.0.0.1:%d" % p for p in ports[:n])
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = [subprocess.Popen([sys.executable, str(script), str(i), hosts, str(steps), str(bound)],
                              env=_env({"DTF_RENDEZVOUS_PORT": str(ports[n])}), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for i in range(n)]
    out = {}
    try:
        for i, p in enumerate(procs):
            text, _ = p.communicate(timeout=240)
            assert p.returncode == 0, text[-3000:]
            line = [l for l in text.splitlines() if l.startswith("RESULT ")][-1]
            out[i] = json.loads(line[len("RESULT "):])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return out


def test_bound_holds_the_fast_worker_back(tmp_path):
    bounded = _run(tmp_path, bound=2)
    # the fast worker may start a step only while clock - min <= 2; the slow one's clock can move during the step
    assert bounded[0]["max_lead"] <= 2, bounded
    assert bounded[0]["max_lead"] >= 1, bounded          # it does use the slack
    assert bounded[1]["max_lead"] <= 0, bounded          # the straggler never leads


def test_unbounded_mode_lets_it_run_ahead(tmp_path):
    free = _run(tmp_path, bound=-1)
    assert free[0]["max_lead"] >= 6, free                # 40 fast steps against one step per 50 ms
