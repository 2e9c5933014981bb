"""Developer check: 512x512 gray frames through k_down512w<1>: which frames / outputs differ from the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
from oracle import oracle as O
lib = L.init(0)
for n in (1, 8, 64, 200):
    gray = synth.frames_gray(n, seed=9, h=512, w=512)
    hgo, qgo = O.hash_frames(gray, num_threads=16)
    for rep in range(3):
        hg, qg = hvd_amd.vpdq.hash_frames(gray)
        badh = np.nonzero((hg != hgo).any(1))[0]
        badq = np.nonzero(qg != qgo)[0]
        bits = np.unpackbits(hg ^ hgo, axis=1).sum(1)
        print(f"n={n} rep={rep}: bad hashes {len(badh)} {badh[:8].tolist()} bits {bits[badh][:8].tolist()}  bad quality {len(badq)} {badq[:8].tolist()} dq {(qg-qgo)[badq][:8].tolist()}", flush=True)
