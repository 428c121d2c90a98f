"""python tools/kernel_meta.py [pattern]: VGPRs / AGPRs / scratch / LDS of every kernel of the rs_*.hip translation units as compiled for gfx950 (hipcc --save-temps in /tmp)."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for tu in ("rs_icp_search", "rs_icp_estimate", "rs_score", "rs_rows"):
    d = tempfile.mkdtemp(prefix="kmeta")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-c", "--save-temps",
                    os.path.join(root, "rescan_amd/csrc/%s.hip" % tu), "-o", "x.o"], cwd=d, check=True, stderr=subprocess.DEVNULL)
    txt = open(os.path.join(d, "%s-hip-amdgcn-amd-amdhsa-gfx950.s" % tu)).read()
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        if pat in name:
            print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
