cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ncc_ties.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
python tests/tools/gpu_ncc_micro.py 2>&1 | grep -v "wide_rows=1"
for o in "wide_rows=2" "wide_rows=3"; do
  echo "== C2 $o"; python bench.py --steps 6 --no-cpu-baseline --opt $o 2>&1 | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('value',d['value'],'single',d['value_single_pair'],'initial_match',s['initial_match'])"
  echo "== C5 $o"; python bench.py --config c5 --steps 4 --no-cpu-baseline --opt $o 2>&1 | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('value',d['value'],'single',d['value_single_pair'],'initial_match',s['initial_match'])"
done
