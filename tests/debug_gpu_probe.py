"""First-contact GPU probe: stage-wise parity of the HIP path against the oracle, printed verbosely."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import synth, capi
from oracle import oracle as orc

w, h = 640, 480
if len(sys.argv) > 1:
    w, h = int(sys.argv[1]), int(sys.argv[2])
levels = 4
K = synth.intrinsics_for(w, h)
(Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(w, h)
print("devices", capi.lib().dvo_amd_device_count(), flush=True)
t = time.time()
pr = capi.RgbdImagePyramid(Ir, Zr, K, levels)
pc = capi.RgbdImagePyramid(Ic, Zc, K, levels)
print("gpu pyramids", time.time() - t, flush=True)
opr = orc.Pyramid(Ir, Zr, K, levels)
opc = orc.Pyramid(Ic, Zc, K, levels)
names = ["I", "Z", "Ix", "Iy", "Zx", "Zy"]
for l in range(levels):
    for p in range(6):
        a = pr.plane(l, p); b = opr.plane(l, p)
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)
        if not same:
            d = np.nanmax(np.abs(a - b)); print("  plane mismatch level", l, names[p], "maxabs", d, "nan pattern equal", np.array_equal(np.isnan(a), np.isnan(b)))
    cnt, mask = pr.select(l)
    rec, idx = opr.select(l)
    om = np.zeros(mask.size, np.uint8); om[idx] = 1
    print("level", l, "select gpu", cnt, "oracle", len(idx), "mask equal", np.array_equal(mask.ravel(), om), flush=True)

trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
for l in range(levels - 1, -1, -1):
    for T in (np.eye(4), Tgt, np.linalg.inv(Tgt)):
        res, nv = trk.residuals(pr, pc, l, T)
        pe, r, valid = orc.compute_residuals(opr, opc, l, T, orc.RCP_EXACT)
        rec, idx = opr.select(l)
        nproc = len(valid)
        oimg = np.full((res.shape[0] * res.shape[1], 2), np.nan, np.float32)
        oimg[idx[:nproc][valid.astype(bool)]] = r
        g = res.reshape(-1, 2)
        eq_nan = np.array_equal(np.isnan(g[:, 0]), np.isnan(oimg[:, 0]))
        both = ~np.isnan(g[:, 0]) & ~np.isnan(oimg[:, 0])
        bit = np.array_equal(g[both].view(np.uint32), oimg[both].view(np.uint32))
        md = np.max(np.abs(g[both] - oimg[both])) if both.any() else 0
        print(f"residuals level {l}: gpu valid {nv} oracle {len(r)} validity-equal {eq_nan} bit-exact {bit} maxdiff {md:.3e} "
              f"mismatch-count {(np.isnan(g[:,0]) != np.isnan(oimg[:,0])).sum()} diffbits {(g[both].view(np.uint32) != oimg[both].view(np.uint32)).sum()}", flush=True)

for name, mode in (("EXACT", orc.RCP_EXACT), ("SSE", orc.RCP_SSE)):
    cfg = orc.default_config(first_level=3, last_level=0, rcp_mode=mode)
    t = time.time(); ro = orc.match(cfg, opr, opc); dto = time.time() - t
    if mode == orc.RCP_EXACT:
        t = time.time(); rg = trk.match(pr, pc); dtg = time.time() - t
        t = time.time(); rg = trk.match(pr, pc); dtg2 = time.time() - t
        print("gpu match time first", dtg, "second", dtg2, "ticks", rg.n_ticks, "passes", rg.n_residual_passes)
        for L in rg.Levels:
            print("  gpu level", L["Id"], "sel", L["ValidPixels"], capi.TERMINATION[L["TerminationCriterion"]], "iters", len(L["Iterations"]),
                  [it["ValidConstraints"] for it in L["Iterations"]], [round(it["TDistributionLogLikelihood"], 1) for it in L["Iterations"]])
    for L in ro["levels"]:
        print("  orc", name, "level", L["id"], "sel", L["valid_pixels"], orc.TERMINATION[L["termination"]], "iters", len(L["iterations"]),
              [it["valid_constraints"] for it in L["iterations"]], [round(it["tdist_loglik"], 1) for it in L["iterations"]])
    print(f"oracle {name}: time {dto:.3f}s  gpu-vs-oracle pose err {synth.pose_error(ro['T'], rg.Transformation):.3e}  "
          f"oracle-vs-gt {synth.pose_error(Tgt, ro['T']):.3e} gpu-vs-gt {synth.pose_error(Tgt, rg.Transformation):.3e}", flush=True)
    if mode == orc.RCP_EXACT:
        itg = rg.Levels[0]["Iterations"][0]; ito = ro["levels"][0]["iterations"][0]
        print("first iteration: P gpu", itg["TDistributionPrecision"].ravel(), "orc", ito["precision"].ravel())
        print("  x gpu", itg["EstimateIncrement"], "\n  x orc", ito["increment"])
        print("  A rel diff", np.max(np.abs(itg["EstimateInformation"] - ito["information"])) / np.max(np.abs(ito["information"])))
        print("info rel diff", np.max(np.abs(rg.Information - ro["information"])) / np.max(np.abs(ro["information"])), "ll", rg.LogLikelihood, ro["loglik"])

# timing
for rep in range(3):
    t = time.time()
    for i in range(20):
        rg = trk.match(pr, pc)
    dt = (time.time() - t) / 20
    print(f"single-pair match: {dt*1e3:.3f} ms  ({1/dt:.1f} pairs/s), ticks {rg.n_ticks}", flush=True)
for B in (4, 16, 64):
    t = time.time()
    out = trk.match_batch([pr] * B, [pc] * B, stats=False)
    dt = time.time() - t
    t = time.time()
    out = trk.match_batch([pr] * B, [pc] * B, stats=False)
    dt = time.time() - t
    e = max(synth.pose_error(o.Transformation, rg.Transformation) for o in out)
    print(f"batch {B}: {dt*1e3:.2f} ms -> {B/dt:.1f} pairs/s, max dev from single {e:.2e}", flush=True)
