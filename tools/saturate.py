"""Keeps the GPU full with 64-frame extraction batches from three threads for <seconds> (default 60): the background load of
`tools/under_load.sh` (the GPU test suite beside it).  usage: python3 tools/saturate.py [seconds]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd.synth import synth_sequence
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
frames = synth_sequence(64)
stop = time.time() + secs
count = [0, 0, 0]
def work(k):
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    while time.time() < stop:
        ex.extract_batch(frames); ex.download_batch(); count[k] += 1
ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
for t in ts: t.start()
for t in ts: t.join()
print("saturate: batches per thread", count, flush=True)
