#!/usr/bin/env python
"""profiles/pmc_traffic.json from the PMC summary of tests/tools/gpu_profile_r06.sh (run on the GPU box), keyed by the hash
of the dominant kernel's source so that bench.py stops quoting it once the kernel changes.  Only a fallback: bench.py
measures FETCH_SIZE / WRITE_SIZE of the dominant kernel itself (roofline.traffic_source = "measured")."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_pmc_main_kernels.csv")
rows = list(csv.DictReader(open(src)))


def get(c, k="void k_refine_skew<4, 1"):  # (prefix: the third template argument is the variant)
    r = [r for r in rows if r["kernel"].startswith(k) and r["counter"] == c][0]
    return float(r["mean"]), int(r["dispatches"])


f, nd = get("FETCH_SIZE")
w, _ = get("WRITE_SIZE")
hit, miss = get("TCC_HIT_sum")[0], get("TCC_MISS_sum")[0]
h = hashlib.sha256()
for fn in ("k_refine_skew.hip", "k_refine.hip", "refine_common.h", "rsm_dev.h"):  # bench.py: kernel_src_sha
    h.update(open(os.path.join(ROOT, "reconstruction_amd", "csrc", fn), "rb").read())
out = {"kernel": "k_refine_skew<4,1>", "workload": "C2_4096x3072_r5_d128", "kernel_src_sha256": h.hexdigest(),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python bench.py "
                 "--no-cpu-baseline --measure-traffic 0 --steps 1 --warmup 0 --inflight 1; tests/tools/gpu_profile_r06.sh, "
                 "summarised by tests/tools/rocpd_pmc.py -> " + os.path.relpath(src, ROOT) + ".  Only the fallback: bench.py "
                 "measures the same two counters itself (roofline.traffic_source = measured)",
       "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w, "dispatches": nd,
       "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section; profiles/pmc_calibration.json measures the same "
                     "factor on known bytes in this kernel's staging pattern); WRITE_SIZE as reported (fp64 stores + the rare path's "
                     "parked registers + the update list)",
       "traffic_bytes_per_launch": (2 * f + w) * 1024.0, "l2_hit_rate": round(hit / (hit + miss), 4)}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
