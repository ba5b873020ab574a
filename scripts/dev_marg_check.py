import importlib, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); abi = importlib.import_module(pkg + ".abi")
buf = importlib.import_module(pkg + ".buffers"); est_m = importlib.import_module(pkg + ".estimator")
import oracle_py
def rel(a, b): return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
for flag, name in ((abi.MARGIN_OLD, "OLD"), (abi.MARGIN_SECOND_NEW, "SECOND_NEW")):
  for tracks, nf, prior in (("sparse", 60, True), ("dense", 150, True), ("sparse", 40, False)):
    opt = abi.default_options(); opt.marginalization_flag = flag
    E = est_m.Estimator(options=opt)
    w = synth.make_windows(3, tracks=tracks, n_feat=nf, max_feat=150, with_prior=prior)
    wg, wo = w.copy(), w.copy()
    E.optimization(wg); pg = E.last_marginalization_info
    po = buf.PriorOutArrays.alloc(3); oracle_py.window_solve(opt, wo, po, buf.summary_alloc(3))
    print(name, tracks, nf, "prior" if prior else "noprior", "n", pg.a["n"], po.a["n"], "nblk", pg.a["nblk"], po.a["nblk"], "marg ms", E.ctx.kernel_ms("marginalize"))
    for i in range(3):
        n = int(po.a["n"][i])
        if n <= 0 or pg.a["n"][i] != n: continue
        nb = int(po.a["nblk"][i])
        Jg, Jo = pg.a["J"][i, :n, :n], po.a["J"][i, :n, :n]
        Hg, Ho = Jg.T @ Jg, Jo.T @ Jo
        d = 1.0 / np.sqrt(np.maximum(np.diag(Ho), 1e-300))
        gg, go = Jg.T @ pg.a["r"][i, :n], Jo.T @ po.a["r"][i, :n]
        print("   kinds eq", bool((pg.a["blk_kind"][i,:nb] == po.a["blk_kind"][i,:nb]).all()), "frames eq", bool((pg.a["blk_frame"][i,:nb] == po.a["blk_frame"][i,:nb]).all()),
              "x0", rel(pg.a["x0"][i,:nb], po.a["x0"][i,:nb]), "H rel", rel(Hg, Ho), "H scaled", rel(Hg*d[:,None]*d[None,:], Ho*d[:,None]*d[None,:]), "g scaled", rel(gg*d, go*d),
              "cost0", 0.5*float(pg.a["r"][i,:n] @ pg.a["r"][i,:n]), 0.5*float(po.a["r"][i,:n] @ po.a["r"][i,:n]))
