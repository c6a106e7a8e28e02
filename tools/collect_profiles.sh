#!/bin/bash
# tools/collect_profiles.sh <tag> (GPU box): everything profiles/<tag>_* is made of, into gpurun_out/<tag>/
#   kernel-trace summary of the bench command, FETCH_SIZE / WRITE_SIZE passes, three SQ-counter passes, kernel-trace
#   summaries of the config-3 and config-5 shapes, the pipe microbenchmarks, the bench line itself.
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
B="--no-cpu-baseline --no-other-configs"
cd $root
tools/prof_kernels.sh ${tag}_fast --precision fast $B --steps 5 --warmup 2 > $out/${tag}_fast_kernel_table.txt 2>&1
cp gpurun_out/${tag}_fast_kernel_stats.csv $out/ 2>/dev/null
echo "kernel trace done"
tools/pmc_kernels.sh ${tag}f FETCH_SIZE $B --steps 1 --warmup 1 > $out/${tag}_fast_pmc_fetch.txt 2>&1
tools/pmc_kernels.sh ${tag}w WRITE_SIZE $B --steps 1 --warmup 1 > $out/${tag}_fast_pmc_write.txt 2>&1
echo "traffic passes done"
{
  echo "# rocprofv3 --pmc passes of bench.py ($tag build), per launch of 4096 utterances: tools/pmc_kernels.sh <tag> \"<counters>\" $B --steps 2 --warmup 1"
  echo "## pass 1"
  tools/pmc_kernels.sh ${tag}s1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" $B --steps 2 --warmup 1
  echo "## pass 2"
  tools/pmc_kernels.sh ${tag}s2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT" $B --steps 2 --warmup 1
  echo "## pass 3"
  tools/pmc_kernels.sh ${tag}s3 "SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" $B --steps 2 --warmup 1
} > $out/${tag}_sq_counters.txt 2>&1
echo "SQ passes done"
for shape in cfg3x256 cfg5x128; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_$shape -o $shape -- python3 $root/tools/prof_shape.py $shape 2 > $out/${tag}_${shape}_run.txt 2>&1 )
  f=$(find $root/gpurun_out/prof_${tag}_$shape -name "*kernel_stats.csv" | head -1)
  cp $f $out/${tag}_${shape%x*}_kernel_stats.csv
done
echo "shape traces done"
tools/run_microbench.sh > /dev/null 2>&1
cp gpurun_out/microbench.txt $out/${tag}_microbench.txt
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_fast.json 2> $out/${tag}_bench_fast.err
echo "bench rc=$?"
ls -la $out
