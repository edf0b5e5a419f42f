#!/bin/bash
# tools/evidence_configs.sh TAG -- on the GPU box: what profiles/ lacked for the configs beside the headline (C2, C3, C5):
# per-kernel rocprofv3 stats, issue counters of the walks, and how far a wave gets into the triangle test.
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R && python3 tools/run_configs.py > $O/${tag}_configs.log 2>&1 || exit 1
for s in import bunny dragon; do
  SCENES=$s bash tools/prof_kernels.sh ${tag}_$s > $O/${tag}_kstats_$s.log 2>&1 || exit 1
  SCENES=$s bash tools/pmc_run.sh ${tag}_issue_$s "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" || exit 1
  python3 tools/pmc_summary.py $O/pmc_${tag}_issue_$s > $O/${tag}_issue_$s.log 2>&1
done
cd $R && python3 tools/tri_stats.py import bunny dragon > $O/${tag}_tri_stats.log 2>&1
