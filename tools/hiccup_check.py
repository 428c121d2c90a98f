"""Where the slow steps of a long bench run come from: step times of the concurrent / serial issue, with and without the live
profile, and with a short GIL switch interval (GPU box)."""
import sys, os, time, gc, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from rescan_amd import capi
torch.cuda.set_device(0); capi.init(0)
w = bench.build_workload(1_000_000, seed=11, knn="hash")
def loop(tag, conc, prof, n=200):
    for _ in range(3): bench.run_step(w, None, conc)
    capi.profile_enable(prof); capi.profile_reset()
    gc.collect(); gc.disable()
    t = [time.perf_counter()]
    for _ in range(n):
        bench.run_step(w, None, conc); t.append(time.perf_counter())
    gc.enable(); capi.profile_enable(False)
    d = np.diff(t) * 1e3; med = np.median(d)
    print(tag, "median %.3f mean %.3f max %.2f slow:" % (med, d.mean(), d.max()), [(k, round(float(v), 1)) for k, v in enumerate(d) if v > 1.3 * med][:12], flush=True)
loop("concurrent+profile" + (" [" + os.environ["HICCUP_TAG"] + "]" if os.environ.get("HICCUP_TAG") else ""), True, True)
if os.environ.get("HICCUP_TAG"):
    sys.exit(0)
loop("concurrent no-profile", True, False)
loop("serial+profile", False, True)
sys.setswitchinterval(1e-4)
loop("concurrent+profile, GIL switch interval 0.1 ms", True, True)
os.environ["HIP_LAUNCH_BLOCKING"] = "0"
