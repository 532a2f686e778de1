export TMPDIR=/tmp RSM_AB_OLD_LIBRARY=1
root=$PWD
for opts in "--opt refine_skew_rps=2" "--opt refine_skew_rps=1"; do
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  out=/tmp/pmc_x; rm -rf $out; mkdir -p $out; cd /tmp
  rocprofv3 --pmc $C --kernel-trace -d $out -o pmc -- python $root/bench.py --pmc-child --inflight 1 --no-cpu-baseline --opt refine_split=0 --opt refine_skew_waves_alone=3840 $opts > $out/log 2>&1
  cd $root
  echo "== [$opts]"
  python tests/tools/rocpd_pmc.py $(find $out -name "*.db") 2>/dev/null | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'k_refine_skew' in r[0] and ', 1>' in r[0] or 'skew2<1>' in r[0]: print('%-36s %-22s %14.0f (%s)' % (r[0][:36], r[1], float(r[3]), r[2]))"
done; done
