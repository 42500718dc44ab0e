import sys, os, ctypes as C
os.environ["DVO_AMD_FIN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
K = synth.intrinsics_for(640, 480)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(640, 480)
pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
for B in (1, 27):
    for rep in range(3):
        trk.match_batch([pr] * B, [pc] * B, stats=False)
        st = (C.c_ulonglong * 8)()
        capi.lib().dvo_amd_debug_finalize_stamps(trk._h, st)
        d = [st[i + 1] - st[i] for i in range(3)]
        print("B", B, "phase cycles: loads+tree", d[0], "outputs", d[1], "publish (tagged pieces, issue only)", d[2], "total", st[3] - st[0])
