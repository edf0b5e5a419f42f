#!/bin/bash
# tools/r05_evidence.sh -- on the GPU box: round 5's profiles in one go (gpurun_out/r05_*): the default bench under rocprofv3
# (kernel stats, HBM traffic, issue counters: tools/collect_profiles.sh), per-kernel stats and issue counters of C5
# (3840x2160 x 16: BASELINE configs[4]), the configs table, and the fuzz tools with every frame's shadow rays through the
# shadow tree, and once more with the primary rays through it as well.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
bash tools/collect_profiles.sh r05 > $O/r05_collect.log 2>&1 || { tail -5 $O/r05_collect.log; exit 1; }
echo "collected"
SIZE=3840x2160 SCENES=dragon bash tools/prof_kernels.sh r05_c5 > $O/r05_kstats_c5.log 2>&1 || exit 1
SIZE=3840x2160 SCENES=dragon bash tools/pmc_run.sh r05_issue_c5 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" || exit 1
python3 tools/pmc_summary.py $O/pmc_r05_issue_c5 > $O/r05_issue_c5.log 2>&1
SIZE=3840x2160 SCENES=dragon bash tools/pmc_run.sh r05_fetch_c5 "FETCH_SIZE" || exit 1
SIZE=3840x2160 SCENES=dragon bash tools/pmc_run.sh r05_write_c5 "WRITE_SIZE" || exit 1
python3 tools/pmc_summary.py $O/pmc_r05_fetch_c5 $O/pmc_r05_write_c5 > $O/r05_traffic_c5.log 2>&1
echo "c5 done"
cd $R && python3 tools/run_configs.py > $O/r05_configs.log 2>&1 || exit 1
CTGPU_SHADOW_TREE=2 timeout -k 10 150 python3 tools/fuzz_parity.py 100 5001 > $O/r05_fuzz_scenes.log 2>&1; echo "fuzz scenes rc $?" >> $O/r05_fuzz_scenes.log
CTGPU_SHADOW_TREE=2 timeout -k 10 150 python3 tools/fuzz_views.py 100 7001 > $O/r05_fuzz_views.log 2>&1; echo "fuzz views rc $?" >> $O/r05_fuzz_views.log
CTGPU_SHADOW_TREE=2 CTGPU_CLOSEST_TREE=2 CTGPU_ORDER=0 timeout -k 10 120 python3 tools/fuzz_parity.py 100 6001 > $O/r05_fuzz_scenes_closest.log 2>&1; echo "fuzz scenes rc $?" >> $O/r05_fuzz_scenes_closest.log
CTGPU_SHADOW_TREE=2 CTGPU_CLOSEST_TREE=2 CTGPU_ORDER=0 timeout -k 10 120 python3 tools/fuzz_views.py 100 8001 > $O/r05_fuzz_views_closest.log 2>&1; echo "fuzz views rc $?" >> $O/r05_fuzz_views_closest.log
tail -q -n 2 $O/r05_fuzz_scenes.log $O/r05_fuzz_views.log $O/r05_fuzz_scenes_closest.log $O/r05_fuzz_views_closest.log
