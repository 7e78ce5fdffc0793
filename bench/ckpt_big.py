import sys, time, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from distributed_tensorflow_amd.models import wide_mlp
from distributed_tensorflow_amd.saver import Saver, list_variables
dev = torch.device("cuda", 0)
m = wide_mlp()
t0 = time.time()
m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=dev, seed=0)
print("compile %.1fs params %d" % (time.time() - t0, m.store.num_params), flush=True)
x = torch.rand(256, 784, device=dev); y = torch.randint(0, 10, (256,), device=dev)
m.train_step(x, y); torch.cuda.synchronize()
s = Saver(m.store, m.optimizer)
t0 = time.time()
prefix = s.save("/tmp/bigck/model.ckpt", global_step=m.global_step)
dt = time.time() - t0
size = sum(os.path.getsize(os.path.join("/tmp/bigck", f)) for f in os.listdir("/tmp/bigck"))
print("save %.1fs %.2f GB -> %.2f GB/s" % (dt, size / 1e9, size / 1e9 / dt), flush=True)
names = dict(list_variables(prefix))
assert names["dense_2/kernel"] == (16384, 16384) and "dense_2/kernel/Adam_1" in names
w = m.store.flat.clone()
m.store.flat.zero_()
t0 = time.time()
s.restore(prefix)
print("restore %.1fs" % (time.time() - t0), flush=True)
assert torch.equal(w, m.store.flat) and m.global_step == 1
print("BIG_CKPT_OK")
