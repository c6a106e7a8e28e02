#!/bin/bash
# tools/collect_profiles.sh <tag> (GPU box): everything profiles/<tag>_* is made of, into gpurun_out/<tag>/
#   kernel-trace summary of the bench command, FETCH_SIZE / WRITE_SIZE passes of the bench command and of the config-3 /
#   config-5 shapes (one counter per pass, no tracing), three SQ-counter passes, kernel-trace summaries of the two
#   shapes, the pipe microbenchmarks, the bench line itself.  Every step is checked: the script stops at the first
#   failure instead of copying stale files of an earlier run (per-tag output directories are removed first).
set -u
tag=${1:-r04}
prec=${2:-fastlin}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
rm -rf $out $root/gpurun_out/prof_${tag}_* $root/gpurun_out/pmc_${tag}*
mkdir -p $out
die() { echo "collect_profiles: $*" >&2; exit 1; }
B="--precision $prec --no-cpu-baseline --no-other-configs"
cd $root
tools/prof_kernels.sh ${tag}_$prec $B --steps 5 --warmup 2 > $out/${tag}_${prec}_kernel_table.txt 2>&1 || die "kernel trace failed"
[ -s gpurun_out/${tag}_${prec}_kernel_stats.csv ] || die "no kernel stats csv"
cp gpurun_out/${tag}_${prec}_kernel_stats.csv $out/
echo "kernel trace done"
tools/pmc_pass.sh ${tag}f FETCH_SIZE $root/bench.py $B --steps 1 --warmup 1 > $out/${tag}_${prec}_pmc_fetch.txt || die "FETCH_SIZE pass failed"
tools/pmc_pass.sh ${tag}w WRITE_SIZE $root/bench.py $B --steps 1 --warmup 1 > $out/${tag}_${prec}_pmc_write.txt || die "WRITE_SIZE pass failed"
echo "traffic passes (config 2) done"
for shape in cfg3x256 cfg5x128; do
  tools/pmc_pass.sh ${tag}f$shape FETCH_SIZE $root/tools/prof_shape.py $shape 1 > $out/${tag}_${shape%x*}_pmc_fetch.txt || die "FETCH_SIZE pass of $shape failed"
  tools/pmc_pass.sh ${tag}w$shape WRITE_SIZE $root/tools/prof_shape.py $shape 1 > $out/${tag}_${shape%x*}_pmc_write.txt || die "WRITE_SIZE pass of $shape failed"
done
echo "traffic passes (config 3, config 5) done"
{
  echo "# rocprofv3 --pmc passes of bench.py ($tag build, $prec), per launch of 4096 utterances: tools/pmc_pass.sh <tag> \"<counters>\" bench.py $B --steps 2 --warmup 1"
  echo "## pass 1"
  tools/pmc_pass.sh ${tag}s1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" $root/bench.py $B --steps 2 --warmup 1 || die "SQ pass 1 failed"
  echo "## pass 2"
  tools/pmc_pass.sh ${tag}s2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT" $root/bench.py $B --steps 2 --warmup 1 || die "SQ pass 2 failed"
  echo "## pass 3"
  tools/pmc_pass.sh ${tag}s3 "SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" $root/bench.py $B --steps 2 --warmup 1 || die "SQ pass 3 failed"
} > $out/${tag}_sq_counters.txt
echo "SQ passes done"
for shape in cfg3x256 cfg5x128; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_$shape -o $shape -- python3 $root/tools/prof_shape.py $shape 2 > $out/${tag}_${shape}_run.txt 2>&1 ) || die "kernel trace of $shape failed"
  f=$(find $root/gpurun_out/prof_${tag}_$shape -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || die "no kernel stats for $shape"
  cp $f $out/${tag}_${shape%x*}_kernel_stats.csv
done
echo "shape traces done"
tools/run_microbench.sh > /dev/null 2>&1
cp gpurun_out/microbench.txt $out/${tag}_microbench.txt
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_$prec.json 2> $out/${tag}_bench_$prec.err
echo "bench rc=$?"
ls -la $out
