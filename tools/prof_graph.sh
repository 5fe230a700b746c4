#!/bin/bash
# usage: tools/prof_graph.sh OUTDIR [workload]   (optionally PT_HOT_LIB=...)  -> OUTDIR/graph_kernel_stats_<workload>.csv
OUT=$1; W=${2:-dimp50}
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_$$ -o g -- python tools/prof_graph_frames.py $W > /dev/null 2> $OUT/prof_graph.err
f=$(find /tmp/pg_$$ -name "*kernel_stats.csv" | head -1)
python - "$f" > $OUT/graph_kernel_stats_$W.csv <<'PY'
import csv, sys
print("Name,Calls,AverageNs,MinNs,MaxNs")
for r in csv.DictReader(open(sys.argv[1])):
    print('"%s",%s,%.1f,%s,%s' % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
head -9 $OUT/graph_kernel_stats_$W.csv
rm -rf /tmp/pg_$$
