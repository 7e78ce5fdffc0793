"""Wall time of the reference's own launch on this machine: `launch --num_ps PS --num_workers W examples/example.py`
(30000 shared global steps of batch 50 by default; every `sess.run` fetches loss, accuracy and the merged summary)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributed_tensorflow_amd import launch  # noqa: E402

ps, w = int(os.environ.get("PS", "1")), int(os.environ.get("WORKERS", "2"))
steps = os.environ.get("LAST_STEP", "30000")
tmp = tempfile.mkdtemp(prefix="dtf_cluster_example_")
env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
env.setdefault("DTF_COMM_BACKEND", "gloo")          # tasks time-sharing the one GPU of a test box
t0 = time.time()
rc = launch.run([os.path.join(ROOT, "examples", "example.py"), "--log_dir", os.path.join(tmp, "logs"), "--seed", "1",
                 "--last_step", steps], num_ps=ps, num_workers=w, max_restarts=0, run_dir=os.path.join(tmp, "run"), env=env,
                log=lambda *_: None)
dt = time.time() - t0
log = open(os.path.join(tmp, "run", "worker_0.attempt0.log")).read()
last = [l for l in log.splitlines() if l.startswith("Epoch:")][-1:]
print("%d ps + %d workers, %s shared global steps of batch 50: %.1f s wall, launcher status %d" % (ps, w, steps, dt, rc))
print("chief's last report: %s" % (last[0] if last else "(none)"))
sys.exit(rc)
