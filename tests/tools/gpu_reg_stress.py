import os, sys, faulthandler, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
faulthandler.enable()
from reconstruction_amd import Context, synth, _lib
lib = _lib.load()
cfg = synth.config_c1()
ctx = Context(0)
ctx.upload_pair(cfg); ctx.run_pair()
for i in range(300):
    n = (1 << 20) + (i % 7) * 4096 * 37
    buf = np.zeros(n, np.uint8)
    assert lib.rsm_host_register(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes)) == 0
    assert lib.rsm_host_unregister(C.c_void_p(buf.ctypes.data)) == 0
    del buf
    r = ctx.download_pair()
    if i % 50 == 0: print("iter", i, r.n_points, flush=True)
print("done")
