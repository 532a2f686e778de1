cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "refine" 2>&1 | tail -4
for o in "refine_defer_to=0" "refine_defer_to=48" "refine_defer_from=3 --opt refine_defer_to=64" "refine_defer_from=6 --opt refine_defer_to=40" "refine_defer_from=2 --opt refine_defer_to=100"; do
  echo "== $o"; python bench.py --steps 6 --no-cpu-baseline --opt $o 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('value',d['value'],'single',d['value_single_pair'],'ms_single',d['ms_single_pair'],'top',s['refine_sweep_top'],'lower',s['refine_sweep'])"
done
