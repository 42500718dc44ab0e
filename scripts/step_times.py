import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
K = synth.intrinsics_for(640, 480)
ref_f = synth.render(640, 480, None, frame_id=0)
curs_f = [synth.render(640, 480, synth.se3_exp(synth.XI_GT_PAIR * (0.6 + 0.1 * i) * (1 if i % 2 == 0 else -1)), frame_id=1 + 2 * i) for i in range(8)]
ref = capi.RgbdImagePyramid(ref_f[0], ref_f[1], K, 4)
curs = [capi.RgbdImagePyramid(f[0], f[1], K, 4) for f in curs_f]
T = 4; B = 256
trks = [capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0)) for _ in range(T)]
refs = [ref] * (B // T); cur = [curs[i % 8] for i in range(B // T)]
def worker(t):
    trks[t].match_batch(refs, cur, stats=False, in_flight=27)
for step in range(12):
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    print(f"step {step}: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
