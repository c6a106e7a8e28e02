#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_decode.sh <tag> [utterances]
# Kernel trace of tools/bench_decode.py; per-kernel table, csv under gpurun_out/.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $root/tools/bench_decode.py "$@" > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $root/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r['Name'][:58].ljust(58), r['Calls'].rjust(4), ('%.3f' % (float(r['AverageNs'])/1e6)).rjust(9), r['Percentage'].rjust(6))
PY
grep '"metric"' $out.log | head -1
