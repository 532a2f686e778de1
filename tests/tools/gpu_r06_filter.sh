#!/bin/bash
# cloud filter on C2's cloud, same box: prebuilt libraries tests/_ab/<name>.so one after the other (bash tests/tools/gpu_r06_filter.sh name1 name2 ...;
# "cur" = the library in the tree), per-call time of rsm_filter_last_cloud + what it decided; then the filter's parity tests with the tree's library
export OMP_NUM_THREADS=16 RSM_AB_OLD_LIBRARY=1
cp reconstruction_amd/librsm_mi355.so /tmp/keep.so
for n in "$@"; do
  [ "$n" = cur ] && cp /tmp/keep.so reconstruction_amd/librsm_mi355.so || cp tests/_ab/$n.so reconstruction_amd/librsm_mi355.so
  echo "[$n]"
  python -u - <<'PY'
import time, torch, sys, hashlib
sys.path.insert(0, '.')
from reconstruction_amd import Context, synth
cfg = synth.config_c2(pair=0)
with Context(0) as ctx:
    ctx.match_pair(cfg, want_cloud=False)
    n = ctx.n_points
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0"); nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    ts = []
    for rep in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    info = ctx.filter_last_info()
    h = hashlib.sha1(rec[:m].cpu().numpy().tobytes() + nrm[:m].cpu().numpy().tobytes()).hexdigest()[:16]
    print("filter ms: min %.2f median %.2f  (%s)" % (min(ts), sorted(ts)[len(ts) // 2], " ".join("%.1f" % t for t in ts)))
    print("kept", m, "stats", st, "info", info, "sha", h)
PY
done
cp /tmp/keep.so reconstruction_amd/librsm_mi355.so
python -m pytest tests/test_gpu_cloud_filter.py -x -q 2>&1 | tail -5
