#!/bin/bash
# kernel trace of the cloud filter on C2's cloud: bash tests/tools/gpu_r05_filter_prof.sh <filter_window value>
export OMP_NUM_THREADS=16
mkdir -p gpurun_out
cat > /tmp/filt_one.py <<'PY'
import time, torch, sys
sys.path.insert(0, '/root/repo')
from reconstruction_amd import Context, synth
w = int(sys.argv[1])
cfg = synth.config_c2(pair=0)
with Context(0) as ctx:
    ctx.match_pair(cfg, want_cloud=False)
    n = ctx.n_points
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda:0"); nrm = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    ctx.set_option("filter_window", w)
    for rep in range(3):
        m, st = ctx.filter_last_cloud(rec.data_ptr(), nrm.data_ptr(), n, 100, 1.0, 2.5, (0.0, 0.0, 0.0))
    print(ctx.filter_last_info())
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/fprof -o f -- python /tmp/filt_one.py $1 > /tmp/fprof.log 2>&1
grep -v rocprofv3 /tmp/fprof.log | tail -3
python /root/repo/tests/tools/rocpd_stats.py $(find /tmp/fprof -name "*.db" | head -1) > /root/repo/gpurun_out/filter_kernel_stats_$1.csv
python - <<PY
import csv
rows = list(csv.DictReader(open('/root/repo/gpurun_out/filter_kernel_stats_$1.csv')))
keys = ('k_sor', 'k_cell', 'k_xq', 'k_cloud_', 'k_gather', 'k_compact', 'k_dist', 'k_keep', 'k_bbox', 'k_finite', 'rocprim', 'k_f64', 'k_pack_f', 'k_sample')
rows = [r for r in rows if any(k in r['Name'] for k in keys)]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:24]:
    tot += float(r['TotalDurationNs'])
    print('%-100s calls %5s total/3 %9.3f ms avg %9.1f us' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 3e6, float(r['AverageNs']) / 1e3))
print('sum of the listed kernels per filter call: %.2f ms' % (tot / 3e6))
PY
