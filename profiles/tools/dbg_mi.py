import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre, synth
counts = synth.generate(300, 250, 17, mode="F")
data, _, _ = pre.normalize(counts, "mi")
n, p = data.shape
eng = fw.Engine("mi", n, p, max_k=3); eng.set_data(data)
net = eng.lgl(feed_forward=False, round_size=0)
print("edges", len(net["edges"]), eng.counters()["cond_tests_ref"])
