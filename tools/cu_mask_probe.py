"""Which CUs does a bit of a HIP CU mask stand for?  Launches a probe grid on a masked stream and prints, per XCD, the set of
(SE, SH, CU) ids the workgroups ran on.  python tools/cu_mask_probe.py [lo hi]  — mask bits [lo, hi) set (default: a few cases)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
from rescan_amd import capi
sys.path.insert(0, os.path.join(ROOT, "tools"))
import benchaux
capi.init(0)
n_cu = torch.cuda.get_device_properties(0).multi_processor_count


def show(name, bits):
    capi.stream_cu_mask(bits)
    xcc, se, sh, cu = benchaux.probe_placement(8192)
    print(f"{name}: {int(sum(bits))} bits set", flush=True)
    for x in sorted(set(xcc.tolist())):
        sel = xcc == x
        ids = sorted(set(zip(se[sel].tolist(), sh[sel].tolist(), cu[sel].tolist())))
        per_se = {}
        for s, h, c in ids:
            per_se.setdefault(s, []).append(c if not h else 16 + c)
        print(f"   xcd {x}: {int(sel.sum()):5d} blocks on {len(ids):2d} CUs  " + "  ".join(f"se{s}:{v}" for s, v in sorted(per_se.items())), flush=True)


if len(sys.argv) == 3:
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    show(f"bits [{lo},{hi})", [1 if lo <= i < hi else 0 for i in range(n_cu)])
else:
    show("all", [1] * n_cu)
    show("bits [0,160)", [1] * 160 + [0] * (n_cu - 160))
    show("bits [160,256)", [0] * 160 + [1] * (n_cu - 160))
    show("bits [0,8)", [1] * 8 + [0] * (n_cu - 8))
    show("bits 0,8,16,..", [1 if i % 8 == 0 else 0 for i in range(n_cu)])
