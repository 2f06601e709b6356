"""Host-to-device upload of 32 MiB of scalars (the host-buffer boundary of commit): pageable copy, pinned copy, and a
chunked pageable -> pinned staging pipeline."""
import time
import numpy as np
import torch

n = 32 << 20
a = np.random.default_rng(1).integers(0, 255, size=n, dtype=np.uint8)
ta = torch.from_numpy(a)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pin.copy_(ta)


def t(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("pageable -> device      %.3f ms" % t(lambda: dev.copy_(ta)))
print("pinned   -> device      %.3f ms" % t(lambda: dev.copy_(pin, non_blocking=True)))
print("pageable -> pinned host %.3f ms (one thread)" % t(lambda: pin.copy_(ta)))
for chunk_mb in (1, 2, 4, 8):
    c = chunk_mb << 20
    stage = [torch.empty(c, dtype=torch.uint8).pin_memory() for _ in range(2)]
    evs = [torch.cuda.Event() for _ in range(2)]

    def pipe():
        for i, off in enumerate(range(0, n, c)):
            k = i & 1
            if i >= 2:
                evs[k].synchronize()
            stage[k].copy_(ta[off:off + c])
            dev[off:off + c].copy_(stage[k], non_blocking=True)
            evs[k].record()

    print("staged pipeline, %d MiB chunks: %.3f ms" % (chunk_mb, t(pipe)))
