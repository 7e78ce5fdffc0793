"""Can the CPU of a process address device memory it has mapped from ANOTHER process's hipIpc handle, and does it see
what a kernel of either process stores there with system scope?  (The doorbell words of the asynchronous mode could
then live in the owners' device regions.)  A fault ends only the child."""
import base64
import ctypes as C
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from distributed_tensorflow_amd.ops import _native as nv  # noqa: E402

lib = nv.kernels()
p, ll = C.c_void_p, C.c_longlong
nv.declare(lib, "dtf_mbox_create", [C.c_int, C.c_int, ll], p)
nv.declare(lib, "dtf_mbox_handle_bytes", [])
nv.declare(lib, "dtf_mbox_local_ptr", [p], p)
nv.declare(lib, "dtf_mbox_local_handle", [p, p])
nv.declare(lib, "dtf_mbox_open_peer", [p, C.c_int, p])
nv.declare(lib, "dtf_mbox_peer_ptr", [p, C.c_int], p)


class Mem:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False), "version": 2, "strides": None}


if len(sys.argv) > 1:      # child: map the parent's region, read it from the CPU, store into it from a kernel
    torch.zeros(1, device="cuda")
    h = lib.dtf_mbox_create(1, 2, 4096)
    raw = base64.b64decode(sys.argv[1])
    rc = lib.dtf_mbox_open_peer(h, 0, C.create_string_buffer(raw, len(raw)))
    ptr = lib.dtf_mbox_peer_ptr(h, 0)
    print("child: open rc %d, peer pointer %#x" % (rc, ptr or 0), flush=True)
    v = C.c_longlong.from_address(ptr + 8).value
    print("child: CPU read of the parent's word 1 through the imported mapping: %d (expected 777)" % v, flush=True)
    t = torch.as_tensor(Mem(ptr, 16), device="cuda")
    t[3] = 31337                                      # a kernel of the child stores into the parent's memory
    torch.cuda.synchronize()
    C.c_longlong.from_address(ptr + 32).value = 99    # and so does the child's CPU
    print("child: done", flush=True)
    sys.exit(0)

torch.zeros(1, device="cuda")
h = lib.dtf_mbox_create(0, 2, 4096)
ptr = lib.dtf_mbox_local_ptr(h)
C.c_longlong.from_address(ptr + 8).value = 777
nb = lib.dtf_mbox_handle_bytes()
buf = C.create_string_buffer(nb)
assert lib.dtf_mbox_local_handle(h, buf) == 0
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
r = subprocess.run([sys.executable, os.path.abspath(__file__), base64.b64encode(bytes(buf.raw)).decode()], capture_output=True,
                   text=True, timeout=120, env=env)
print(r.stdout.strip())
print("child exit status %d" % r.returncode)
print("parent: CPU reads word 3 (stored by the child's kernel): %d (expected 31337), word 4 (child's CPU): %d (expected 99)"
      % (C.c_longlong.from_address(ptr + 24).value, C.c_longlong.from_address(ptr + 32).value))
