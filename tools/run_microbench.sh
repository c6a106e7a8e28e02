#!/bin/bash
# tools/run_microbench.sh (GPU box): builds and runs the four pipe microbenchmarks that DESIGN.md section 6 argues from
# (fp64 MFMA and VALU time add on gfx950; the fp64 vector peak; MFMA occupancy), output to gpurun_out/microbench.txt.
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/microbench.txt
: > $out
for t in mfma_valu_coexec mfma_valu_samewave valu_f64_peak mfma_f64_occupancy mfma_lds_loop; do
  echo "== tools/$t.hip" >> $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/$t $root/tools/$t.hip >> $out 2>&1 && /tmp/$t >> $out 2>&1
done
cat $out
