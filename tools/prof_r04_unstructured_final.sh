#!/bin/bash
# End of round 4: HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes, serial mode) of the step kernels on
# the unstructured operators of the bench, ONE operator per pass so kernel names do not mix: uniformly random columns, power-law rows
# (column-blocked form), a shuffled 2-D grid and a triangulated planar mesh numbered at random (both: mesh patches -> patch form).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r04_pmc_traffic_unstructured_final.txt
for K in rand5 powerlaw shuf_grid shuf_trimesh; do
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/uf_${K}_$i -o c -- python tools/general_sparse.py $K > gpurun_out/uf_${K}_$i.log 2>&1
  done
  {
    echo "== $K =="
    grep "^{" gpurun_out/uf_${K}_1.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   path', d['path'], ' contract per step %.1f MB' % (d['alg_GB'] * 1e3 / 30), ' (serial mode, under the counter pass: %.3f of the contract)' % d['frac'])"
    python tools/pmc_summary.py gpurun_out/uf_${K}_1/c_counter_collection.csv gpurun_out/uf_${K}_2/c_counter_collection.csv gpurun_out/uf_${K}.json | grep -E "k_pipe|k_fused|k_update2|k_spmv|k_cbf|k_gather|k_combine1"
    echo
  } >> gpurun_out/r04_pmc_traffic_unstructured_final.txt
done
find gpurun_out -name "*.db" -delete
cat gpurun_out/r04_pmc_traffic_unstructured_final.txt
