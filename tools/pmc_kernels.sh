#!/bin/bash
# usage (GPU box): tools/pmc_kernels.sh <tag> "<counters>" <bench.py args...>
# One PMC pass (no other tracing) over a short bench run; prints per-kernel counter means.
tag=$1; shift; ctrs=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o $tag -- python3 $root/bench.py "$@" > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    if not any(x in k for x in ('fused','k_lin_z','k_dp','k_post','k_exp_rows','k_atb','mfma','k_xi','k_reduce','k_pframe','k_ztf','k_mass')): continue
    print(k, 'launches', len(n[k]))
    for c,v in sorted(acc[k].items()): print('   %-32s %.4g' % (c, v/len(n[k])))
PY
