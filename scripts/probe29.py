import sys; sys.path.insert(0, '/root/repo')
from jellyfish_amd import lib as L
L.load_library(); L.init(0)
print("mad/s %.3e" % L.probe(0))
for f in range(4):
    a = L.probe(2, f); b = L.probe(3, f)
    print("field %d  mul/s %.3e   mul29/s %.3e  ratio %.2f" % (f, a, b, b / a))
