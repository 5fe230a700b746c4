#!/bin/bash
# rocprofv3 PMC passes over the IoU refinement (device in / device out route: nothing polls host memory, which does not
# finish under the serialised dispatch of counter collection) and the ATOM optimisers; one counter group per pass,
# kernel-trace only, each pass under its own timeout; per-kernel averages -> gpurun_out/TAG/{iou,atom}_pmc_*.txt
TAG=${1:-r04_iou_atom_pmc}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
cat > $OUT/iou_device_route.py <<'PY'
import sys, os, types, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pytracking_amd import synth, iou_refine as IR
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_iou
dev = torch.device("cuda", 0)
net = bench_iou.Net().to(dev).eval() if hasattr(bench_iou, "Net") else None
c3, c4, m3, m4, boxes = synth.iou_inputs(7302)
T = lambda a: torch.from_numpy(a).to(dev)
if net is None:
    raise SystemExit("bench_iou.Net not found")
b = T(boxes)
for _ in range(12):
    IR.refine_boxes(net, (T(m3), T(m4)), (T(c3), T(c4)), b, 5, 1.0, 1.0, False)
torch.cuda.synchronize()
PY
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pi$i -- python $OUT/iou_device_route.py > $OUT/pi$i.log 2>&1
  echo "iou pass $i exit $?" >> $OUT/passes.log
  python tools/pmc_summary.py $OUT/pi$i > $OUT/iou_pmc_$i.txt; rm -rf $OUT/pi$i
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pa$i -- python tools/bench_atom.py > $OUT/pa$i.log 2>&1
  echo "atom pass $i exit $?" >> $OUT/passes.log
  python tools/pmc_summary.py $OUT/pa$i > $OUT/atom_pmc_$i.txt; rm -rf $OUT/pa$i
done
cat $OUT/passes.log
