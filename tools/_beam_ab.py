import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from taiyaki_amd import decodeutil, synth
T, N = 2000, 512
sc = torch.from_numpy((synth.scores(T, N, 40, 5) * np.float32(0.8)).astype(np.float32)).cuda()
for guided in (True, False):
    for width in (5, 2, 1):
        decodeutil.beamsearch(sc, 0.0, width, guided); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            decodeutil.beamsearch(sc, 0.0, width, guided)
        torch.cuda.synchronize()
        print("guided", guided, "width", width, "%.2f ms" % ((time.time() - t0) / 5 * 1e3), flush=True)
