#!/bin/bash
# usage (GPU box): tools/prof_any.sh <tag> <python script + args...>   kernel trace + per-kernel table
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 "$@" > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $root/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(4), ('%.3f' % (float(r['AverageNs'])/1e6)).rjust(9), r['Percentage'].rjust(6))
PY
tail -2 $out.log
