#!/bin/bash
# usage: tools/gpu_pmc4.sh TAG   -- round-4 counter evidence for the solver kernels of BOTH frame workloads (DiMP-50, PrDiMP-50):
# one rocprofv3 --pmc pass per counter group (kernel-trace only, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit
# one pass), each over a short EAGER bench.py run without the side workloads (they poll pinned host memory; under serialised
# dispatch that run did not finish in round 3) and under its own `timeout`.  Per-kernel averages -> gpurun_out/TAG/<workload>_<group>.txt,
# all of them concatenated -> gpurun_out/TAG/pmc_counters.txt; tools/pmc_traffic.py turns that file into profiles/pmc_traffic.json.
TAG=${1:-r04_pmc}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
run_pass() {   # workload, group name, counters...
  local W=$1 G=$2; shift 2
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/p_${W}_$G -- \
      python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-graph --no-other \
      > $OUT/p_${W}_$G.json 2> $OUT/p_${W}_$G.err
  echo "pass $W $G exit $?" >> $OUT/passes.log
  python tools/pmc_summary.py $OUT/p_${W}_$G > $OUT/${W}_$G.txt
  rm -rf $OUT/p_${W}_$G
}
for W in dimp50 prdimp50; do
  run_pass $W fetch FETCH_SIZE
  run_pass $W write WRITE_SIZE
  run_pass $W mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
done
run_pass dimp50 mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
run_pass dimp50 wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
for f in $OUT/dimp50_*.txt $OUT/prdimp50_*.txt; do
  echo "=== $(basename $f .txt) ==="; grep -v -E "rocclr|at::native" $f | grep -A8 -E "^(void )?(\(anonymous namespace\)::)?k_(corr2|adj2|fast_)"
done > $OUT/pmc_counters.txt
cat $OUT/passes.log
