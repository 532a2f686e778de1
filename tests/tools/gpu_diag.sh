#!/bin/bash
# one-shot diagnostics for the GPU box (kept: documents how the GPU logs under profiles/ were produced)
mkdir -p gpurun_out
{
echo "== cpu"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print('affinity',len(os.sched_getaffinity(0)))"
echo "== oracle timing"; 
for t in 8 32 256; do OMP_NUM_THREADS=$t python - <<'PY'
import time,sys,os
sys.path.insert(0,'.')
from oracle import oracle as orc
from reconstruction_amd import synth
cfg=synth.config_small(160,96,2,radius=2)
t=time.time(); orc.match_pair(cfg); print('threads',os.environ['OMP_NUM_THREADS'],'oracle 160x96: %.2fs'%(time.time()-t))
PY
done
} > gpurun_out/diag.log 2>&1
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
timeout ${1:-600} python -u -m pytest tests -m gpu -v --durations=15 -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -5 gpurun_out/diag.log; tail -40 gpurun_out/pytest.log
