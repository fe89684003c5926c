import sys, time
sys.path.insert(0, ".")
import numpy as np
import caf_cookoff_amd as caf
from caf_cookoff_amd.synth import make_batch
import bench
eng = caf.Engine(0)
fr = caf.bench_shifts()
plan = eng.plan(4096, fr, 48000)
nd, hs, lags, _ = make_batch(64, 4096, 48000, seed0=5000)
for nslots, batch, split in ((2,1,False),(3,1,False),(4,1,False),(5,1,False),(6,1,False),(8,1,False),(2,2,True),(3,2,True),(2,4,True),(3,4,True),(2,8,True)):
    v, us, ok = bench.stream_run(plan, nd, hs, lags, 1000, nslots, batch, split)
    print(f"slots={nslots} batch={batch} split={split}: {v:8.0f} surfaces/s ({us:.1f} us/surface) ok {ok}")
