set -x
cd /root/repo
timeout 900 python -m pytest tests/test_beamsearch.py -x -q -m gpu 2>&1 | tail -15
timeout 300 python - <<'PY'
import torch, time, numpy as np
from taiyaki_amd import decodeutil, synth
for T, N in ((2000, 512), (4000, 1024), (800, 128)):
    sc = torch.from_numpy((synth.scores(T, N, 40, 5) * np.float32(0.8)).astype(np.float32)).cuda()
    decodeutil.beamsearch(sc, 0.0, 5, True); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): decodeutil.beamsearch(sc, 0.0, 5, True)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print("beam T=%d N=%d: %.2f ms/call, %.2f Mblocks/s" % (T, N, dt * 1e3, T * N / dt / 1e6))
PY
