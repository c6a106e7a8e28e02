#!/bin/bash
# usage (GPU box): tools/pmc_any.sh <tag> "<counters>" <python script + args...>   one PMC pass, per-kernel means
tag=$1; shift; ctrs=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o $tag -- python3 "$@" > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:48]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    if not k.startswith(('void k_','k_')): continue
    print(k, 'launches', len(n[k]))
    for c,v in sorted(acc[k].items()): print('   %-36s %.4g' % (c, v/len(n[k])))
PY
