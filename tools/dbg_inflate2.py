import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, corpus, oracle_lib as o, rsx_pkg
rsx = rsx_pkg.load()
eng = rsx.Engine(0)
t = corpus.text(40000, 5)
c = o.compress(t, 6, 0)
out, st = eng.batch_decompress([c], [len(t)+64], 0)
got = out[0]
a = np.frombuffer(got, dtype=np.uint8); b = np.frombuffer(t, dtype=np.uint8)
d = np.nonzero(a != b)[0]
print("mismatch", d)
p = int(d[0])
print("want", t[p-8:p+12], "got", got[p-8:p+12])
# where does the wanted text occur earlier
w = t[p:p+3]
print("occurrences of wanted 3 bytes before p:", [i for i in range(p) if t[i:i+3] == w][-5:])
g3 = got[p:p+3]
print("occurrences of got 3 bytes before p:", [i for i in range(p) if t[i:i+3] == g3][-5:])
