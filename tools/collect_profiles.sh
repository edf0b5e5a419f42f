#!/bin/bash
# tools/collect_profiles.sh TAG -- on the GPU box: rocprofv3 passes over the default bench.py run
# (kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and cache counters in passes of their own, as
# MI355X_MICROARCH.md prescribes), summarised into gpurun_out/TAG_summary.{md,json}.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${tag}_stats.log 2>&1
rm -rf $R/gpurun_out/${tag}_fetch
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${tag}_fetch.log 2>&1
rm -rf $R/gpurun_out/${tag}_write
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${tag}_write.log 2>&1
rm -rf $R/gpurun_out/${tag}_cache
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/${tag}_cache -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${tag}_cache.log 2>&1
rm -rf $R/gpurun_out/${tag}_sq
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/${tag}_sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${tag}_sq.log 2>&1
rm -rf $R/gpurun_out/${tag}_sqc
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ --output-format csv -d $R/gpurun_out/${tag}_sqc -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${tag}_sqc.log 2>&1
rm -rf $R/gpurun_out/${tag}_sq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/${tag}_sq2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $R/gpurun_out/${tag}_sq2.log 2>&1
cd $R && python3 tools/summarise_profiles.py $tag
# the default command itself (three frames in flight): bench.py's event timings of the timed region must agree with these
cd /tmp
rm -rf $R/gpurun_out/${tag}_stats_default
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats_default -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${tag}_stats_default.log 2>&1
cd $R && { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (the default command: several frames in flight, see frames_in_flight in the bench line)"; echo; echo '```'; python3 tools/kstats.py gpurun_out/${tag}_stats_default "kernel stats"; echo '```'; echo; echo "bench line of that run:"; echo; echo '```'; grep '^{' gpurun_out/${tag}_stats_default.log; echo '```'; } > gpurun_out/${tag}_stats_default.md
