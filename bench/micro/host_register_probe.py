"""Can this process's GPU address a POSIX shared-memory segment?  Calls hipHostRegister of the HIP runtime on segments
of a few sizes (and on an anonymous page-aligned buffer for comparison) and prints the return codes."""
import ctypes as C
import mmap
from multiprocessing import shared_memory

import torch

torch.zeros(1, device="cuda")                       # the runtime is up
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
PORTABLE, MAPPED = 1, 2
for size in (1152, 4096, 65536):
    shm = shared_memory.SharedMemory(create=True, size=size)
    base = C.addressof(C.c_char.from_buffer(shm.buf))
    for nbytes in sorted({size, (size + 4095) // 4096 * 4096}):
        rc = hip.hipHostRegister(C.c_void_p(base), nbytes, PORTABLE | MAPPED)
        print("shared-memory segment of %6d bytes: hipHostRegister(%6d bytes) -> %d" % (size, nbytes, rc), flush=True)
        if rc == 0:
            hip.hipHostUnregister(C.c_void_p(base))
    shm.close()
    shm.unlink()
anon = mmap.mmap(-1, 65536)
base = C.addressof(C.c_char.from_buffer(anon))
rc = hip.hipHostRegister(C.c_void_p(base), 65536, PORTABLE | MAPPED)
print("anonymous private mapping of  65536 bytes: hipHostRegister -> %d" % rc, flush=True)
if rc == 0:
    hip.hipHostUnregister(C.c_void_p(base))
