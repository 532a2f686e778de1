#!/bin/bash
# Development aid: same-box A/B of the working tree (B) against the committed HEAD sources (A, exported to
# tests/_ab/csrc by the caller: git archive HEAD reconstruction_amd/csrc | tar -x -C tests/_ab --strip-components=1).
export OMP_NUM_THREADS=16 OMP_WAIT_POLICY=passive
run() { for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$1', d['ms_per_step'], round(s['refine_sweep_top']+s['refine_sweep'],2))"; done; }
run B
mkdir -p /tmp/B && cp reconstruction_amd/csrc/*.hip reconstruction_amd/csrc/*.h reconstruction_amd/csrc/*.cpp /tmp/B/
cp tests/_ab/csrc/*.hip tests/_ab/csrc/*.h tests/_ab/csrc/*.cpp reconstruction_amd/csrc/
make -C reconstruction_amd/csrc > /dev/null 2>&1; run A
cp /tmp/B/* reconstruction_amd/csrc/
make -C reconstruction_amd/csrc > /dev/null 2>&1; run B
