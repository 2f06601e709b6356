"""Developer probe: random 64-byte gather rate vs table size (the bucket accumulation's access pattern)."""
import sys; sys.path.insert(0, '/root/repo')
from jellyfish_amd import lib as L
L.load_library(); L.init(0)
for gib in (1, 4, 12, 24):
    a = L.probe(4, gib); b = L.probe(5, gib)
    print("table %2d GiB  dependent %.3e gathers/s (2^24 x 12 in %.2f ms)   independent %.3e (%.2f ms)"
          % (gib, a, 201326592 / a * 1e3, b, 201326592 / b * 1e3), flush=True)
