"""How fast does a finished frame reach pageable host memory?  (a) copy from a pinned staging buffer by the CPU, (b) register the
pageable buffer and let the GPU write into it."""
import time, ctypes as C
import numpy as np, torch
n = 1920 * 1080 * 3
dev = torch.zeros(n, dtype=torch.float32, device="cuda")
pinned = torch.empty(n, dtype=torch.float32, pin_memory=True)
dst = [np.empty(n, np.float32) for _ in range(8)]
for d in dst: d[:] = 1.0  # touch the pages
pn = pinned.numpy()
t0 = time.perf_counter()
for d in dst: np.copyto(d, pn)
print("cpu copy pinned -> pageable: %.2f ms per 25 MB frame" % ((time.perf_counter() - t0) / len(dst) * 1e3))
rt = torch.cuda.cudart()
for rep in range(2):
    t0 = time.perf_counter()
    if any(int(rt.cudaHostRegister(d.ctypes.data, d.nbytes, 0)) != 0 for d in dst[:1]):
        print("hipHostRegister refuses this memory on this box (measured: it does): only the staged copy is available")
        break
    for d in dst[1:]: assert int(rt.cudaHostRegister(d.ctypes.data, d.nbytes, 0)) == 0
    t1 = time.perf_counter()
    for d in dst:
        t = torch.from_numpy(d)
        t.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for d in dst: assert int(rt.cudaHostUnregister(d.ctypes.data)) == 0
    t3 = time.perf_counter()
    print("register %.2f ms, copy %.2f ms, unregister %.2f ms per frame" % ((t1 - t0) / 8e-3, (t2 - t1) / 8e-3, (t3 - t2) / 8e-3))
t0 = time.perf_counter()
for d in dst: torch.from_numpy(d).copy_(dev)
torch.cuda.synchronize()
print("blocking copy into pageable memory: %.2f ms per frame" % ((time.perf_counter() - t0) / 8e-3))
