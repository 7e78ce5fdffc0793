"""Can this process's GPU address a POSIX shared-memory segment (hipHostRegister)?  Prints the return codes for a few
sizes and alignments, and checks that a GPU store into the registered words is seen by the CPU."""
import ctypes as C
import os
import sys
from multiprocessing import shared_memory

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import _native as nv  # noqa: E402

lib = nv.kernels()
nv.declare(lib, "dtf_host_register", [C.c_void_p, C.c_longlong, C.c_void_p])
nv.declare(lib, "dtf_host_unregister", [C.c_void_p])
torch.cuda.init()
torch.zeros(1, device="cuda")
for size in (1152, 4096, 65536):
    shm = shared_memory.SharedMemory(create=True, size=size)
    base = C.addressof(C.c_char.from_buffer(shm.buf))
    for nbytes in (size, (size + 4095) // 4096 * 4096):
        out = C.c_void_p()
        rc = lib.dtf_host_register(C.c_void_p(base), nbytes, C.byref(out))
        print("segment %6d bytes (mapping at %#x, mapped size %d): register %6d bytes -> rc %d, device alias %s"
              % (size, base, shm.size, nbytes, rc, hex(out.value) if out.value else None), flush=True)
        if rc == 0:
            words = np.frombuffer(shm.buf, dtype=np.uint64, count=size // 8)
            t = torch.as_tensor(np.zeros(1), device="cuda")
            lib.dtf_host_unregister(C.c_void_p(base))
            del words
    shm.close()
    shm.unlink()
