import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rnnoise_amd import capi, synth
blob = bench.load_blob()
with capi.instrumented():
    m = capi.Model(blob)
    pcm = synth.batch_pcm([3, 8], 10, lead_silence=2)
    b = capi.Batch(m, 2)
    b.debug_pitch(arm_only=True)
    rows = []
    for t in range(10):
        b.process(pcm[t:t + 1])
        taps = b.debug_pitch()
        rows.append(taps[0, 864:874].copy())
    np.save(sys.argv[1], np.array(rows))
    print(np.array(rows)[5:9])
