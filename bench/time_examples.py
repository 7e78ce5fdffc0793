"""Wall time of the example scripts on one GPU (single machine): the reference-style session loop of example.py
(30000 steps of batch 50, loss / accuracy / summaries fetched every step) and the Keras-style fit of example2.py."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "r2")
os.makedirs(out, exist_ok=True)
for name, args in (("example.py", ["--last_step", "30000", "--seed", "1"]), ("example2.py", ["--seed", "1"])):
    log = os.path.join(out, name.replace(".py", "_single.log"))
    t0 = time.time()
    with open(log, "w") as f:
        rc = subprocess.run([sys.executable, os.path.join(ROOT, "examples", name), "--log_dir", "/tmp/dtf_" + name] + args,
                            stdout=f, stderr=subprocess.STDOUT, timeout=600).returncode
    dt = time.time() - t0
    tail = open(log).read().strip().splitlines()[-3:]
    print("%s rc=%d wall %.1f s" % (name, rc, dt))
    for l in tail:
        print("   ", l[:200])
