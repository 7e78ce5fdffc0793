"""Is device memory (hipMalloc) addressable by the CPU of this process (large BAR)?  A child process tries to read and
write it through the pointer; a fault there ends only the child."""
import ctypes as C
import os
import subprocess
import sys

CHILD = r'''
import ctypes as C, sys, torch
t = torch.arange(16, dtype=torch.int64, device="cuda") + 100
torch.cuda.synchronize()
p = t.data_ptr()
print("device pointer %#x" % p, flush=True)
v = C.c_longlong.from_address(p + 8).value            # faults unless the CPU can address device memory
print("CPU read of element 1: %d (expected 101)" % v, flush=True)
C.c_longlong.from_address(p + 16).value = 4242
torch.cuda.synchronize()
print("GPU sees the CPU's store: %d (expected 4242)" % int(t[2].item()), flush=True)
'''
r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=120)
print(r.stdout.strip())
print("child exit status %d%s" % (r.returncode, " (signal %d)" % -r.returncode if r.returncode < 0 else ""))
