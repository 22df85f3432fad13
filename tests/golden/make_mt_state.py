#!/usr/bin/env python3
"""Writes tests/golden/mt19937_state_c3.npz: the raw MT19937 (np.random.seed(0)) generator state at the output block
that holds the first path of the LAST image column of BASELINE config C3 (4096x4096, S=256): path 17 175 674 880,
block 110 100 479.  Walking there takes 1.1e8 sequential twists (~2 minutes of one core), which no test should repeat:
the state is 2496 bytes.  tests/test_gpu_parity.py renders that column through the reference's exact pipeline from it.
The state is cross-checked below against an independent implementation (numpy's own MT19937 bit generator, advanced by
drawing) on a short prefix, and the oracle's C implementation on the same block."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ascendpathtracing_amd import gen_data

W = H = 4096
S = 256
first_path = (W * H - H) * 4 * S
block = first_path // 156
# cross-check of the convention on a short walk: numpy's legacy generator, state after drawing 156*(b) paths = 624*b words
b_small = 1000
rs = np.random.RandomState(0)
rs.random_sample(2 * 156 * b_small)                     # 2 doubles per path, 2 words per double
key = rs.get_state()[1].astype(np.uint32)               # numpy twists lazily: after exactly 624*b words pos == 624 and
assert rs.get_state()[2] == 624                         # the key is the state that produced block b-1 ...
st, _ = gen_data.mt19937_checkpoints_window(b_small - 1, 1, 0, 8, None)
assert np.array_equal(st[0], key), "raw-state convention differs from numpy's"
t0 = time.time()
st, _ = gen_data.mt19937_checkpoints_window(block, 1, 0, 8, None)
print(f"walked to block {block} in {time.time() - t0:.0f} s")
np.savez(os.path.join(ROOT, "tests", "golden", "mt19937_state_c3.npz"), block=np.uint64(block), first_path=np.uint64(first_path),
         state=st[0], width=W, height=H, samples=S, seed=0)
print("wrote tests/golden/mt19937_state_c3.npz")
