#!/bin/bash
# usage (GPU box): tools/pmc_pass.sh <tag> "<counters>" <python script + args...>
# ONE rocprofv3 --pmc pass (no tracing) over a python command; prints per-kernel counter means.  The first output line is
# the fingerprint of the kernel sources AT MEASUREMENT TIME (bench.kernels_fingerprint()): tools/make_traffic_json.py
# refuses passes whose fingerprints differ or are missing.  Fails loudly: stale output of an earlier pass is removed
# first, a profiler error or an empty result is a non-zero exit.
set -u
tag=$1; shift; ctrs=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
rm -rf "$out" "$out.log"
echo "# kernels_sha16 $(cd $root && python3 -c 'import bench; print(bench.kernels_fingerprint())')"
echo "# counters: $ctrs ; command: python3 $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o $tag -- python3 "$@" > $out.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then echo "pmc_pass $tag: rocprofv3 exited $rc (see $out.log)" >&2; tail -5 $out.log >&2; exit 1; fi
f=$(find $out -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then echo "pmc_pass $tag: no counter_collection.csv under $out" >&2; exit 1; fi
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:64]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
if not acc: sys.exit("empty counter file")
for k in acc:
    if not k.startswith(('void k_','k_')): continue
    print(k, 'launches', len(n[k]))
    for c,v in sorted(acc[k].items()): print('   %-36s %.6g' % (c, v/len(n[k])))
PY
