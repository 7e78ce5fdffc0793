"""The reference's topology - ps tasks own the variables, workers push gradients and pull parameters every step
(/root/reference/example.py:128-139,168-170) - on the two data planes of the asynchronous mode: peer regions +
doorbell words, and torch.distributed send/recv with the folded control traffic.  Starts PS ps tasks + WORKERS workers
as processes on DEVICE (default cuda:0: all on one GPU), trains the reference network (64-128-128-32, batch 50, Adam)
for STEPS steps per worker and prints steps/s per worker and control-plane calls per step.

    python bench/async_ps_planes.py            # both planes, 1 ps + 2 workers
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TASK = r'''
import json, os, sys, time, torch
sys.path.insert(0, %(root)r)
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.models import xor_mlp
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

job, index, ps_hosts, w_hosts, steps, device = sys.argv[1], int(sys.argv[2]), sys.argv[3].split(","), sys.argv[4].split(","), int(sys.argv[5]), sys.argv[6]
srv = Server(ClusterSpec({"ps": ps_hosts, "worker": w_hosts}), job, index, config={"device": device})
if job == "ps":
    srv.join()
    sys.exit(0)
ms = os.environ.get("MAX_STALENESS", "")
strat = AsyncPSStrategy(srv, max_staleness=int(ms) if ms else None)
m = xor_mlp()
m.compile(loss="mean_squared_error", optimizer="adam", metrics=["accuracy"], strategy=strat, device=device, seed=index)
g = torch.Generator().manual_seed(index)
x = torch.randint(0, 2, (50 * 64, 64), generator=g).float().to(device)
y = torch.cat([(x[:, :32] != x[:, 32:]).float()], 1).to(device)
step = m.train_step_graphed if (device != "cpu" and os.environ.get("GRAPHED", "0") == "1") else m.train_step
def run(n):
    for i in range(n):
        lo = (i %% 64) * 50
        step(x[lo:lo + 50], y[lo:lo + 50])
    if device != "cpu":
        torch.cuda.synchronize()
run(100)
strat.cp0.barrier("warm", len(w_hosts), timeout_s=120)
c0, t0 = strat.control_calls(), time.perf_counter()
if os.environ.get("PROFILE") == "1" and index == 0:
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    run(steps)
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
    sys.stderr.write(out.getvalue())
else:
    run(steps)
dt = time.perf_counter() - t0
calls = (strat.control_calls() - c0) / steps
strat.worker_done()
strat.cp0.barrier("done", len(w_hosts), timeout_s=120)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"worker": index, "plane": strat.plane, "steps_per_s": steps / dt, "us_per_step": dt / steps * 1e6,
                              "control_calls_per_step": calls, "loss": m.read_stats()["loss"]}))
'''


def ports(n):
    socks = [socket.socket() for _ in range(n)]
    for s in socks:
        s.bind(("127.0.0.1", 0))
    out = [s.getsockname()[1] for s in socks]
    for s in socks:
        s.close()
    return out


def main():
    n_ps, n_w = int(os.environ.get("PS", "1")), int(os.environ.get("WORKERS", "2"))
    steps, device = int(os.environ.get("STEPS", "2000")), os.environ.get("DEVICE", "cuda:0")
    script = "/tmp/dtf_async_ps_planes_task.py"
    with open(script, "w") as f:
        f.write(TASK % {"root": ROOT})
    for plane in os.environ.get("PLANES", "region,p2p").split(","):
        pp = ports(n_ps + n_w + 1)
        ps_hosts = ",".join("127.0.0.1:%d" % p for p in pp[:n_ps])
        w_hosts = ",".join("127.0.0.1:%d" % p for p in pp[n_ps:n_ps + n_w])
        env = dict(os.environ, DTF_RENDEZVOUS_PORT=str(pp[-1]), DTF_PS_DATA_PLANE=plane, DTF_COMM_BACKEND="gloo",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS"):
            env.pop(k, None)
        tasks = [("ps", j) for j in range(n_ps)] + [("worker", i) for i in range(n_w)]
        procs = [subprocess.Popen([sys.executable, script, job, str(i), ps_hosts, w_hosts, str(steps), device], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for job, i in tasks]
        try:
            outs = [p.communicate(timeout=300)[0] for p in procs]
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        rows = []
        if os.environ.get("PROFILE") == "1":
            print(outs[n_ps][-6000:])
        for (job, i), p, out in zip(tasks, procs, outs):
            if p.returncode != 0:
                print("== %s %d rc=%s\n%s" % (job, i, p.returncode, out[-2000:]))
                return 1
            rows += [json.loads(l[7:]) for l in out.splitlines() if l.startswith("RESULT ")]
        total = sum(r["steps_per_s"] for r in rows)
        print("%d ps + %d workers on %s, batch 50, %s, plane %-6s: %.0f steps/s in total (%.0f us per step per worker), "
              "%.2f control-plane calls per step" % (n_ps, n_w, device,
                                                     "graph replay" if os.environ.get("GRAPHED", "0") == "1" else "eager launches",
                                                     rows[0]["plane"], total,
                                                     sum(r["us_per_step"] for r in rows) / len(rows),
                                                     sum(r["control_calls_per_step"] for r in rows) / len(rows)), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
