import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import oracle
from labrador_ldpc_amd import LDPCCode, device_count
print("devices", device_count())
rng = np.random.default_rng(1)
for code in LDPCCode:
    for dt in (np.float32, np.int8):
        llrs,_ = oracle.awgn_llrs(code, rng, 64, 2.5, dt)
        t=time.time(); og, ig, sg = code.decode_ms_batch(llrs, 25); tg=time.time()-t
        oc, ic, sc, _ = oracle.decode_ms_batch(code, llrs, 25)
        bad = int(((ig!=ic)|(sg!=sc)|(og!=oc).any(axis=1)).sum())
        print(code.name, np.dtype(dt).name, "bad", bad, "mean it", ic.mean(), "succ", sc.mean(), "gpu %.3fs"%tg, flush=True)
