#!/bin/bash
# The measuring suite of a round, run on the GPU box in ONE gpurun call (a box costs minutes to get):
#   bash tools/round_profile.sh r03 [quick]
# Every step runs under its own timeout; a step that is killed or times out ends the script (no GPU step behind a hung one).
set -u
R=${1:-rXX}; OUT=gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp
step() {  # name, seconds, command...
    local name=$1 secs=$2; shift 2
    echo "== $name" | tee -a $OUT/steps.log
    timeout -k 10 $secs "$@"; local rc=$?
    echo "== $name rc=$rc" | tee -a $OUT/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping" | tee -a $OUT/steps.log; exit 1; fi
    return $rc
}
step pytest 900 bash -c "python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log"
step text 120 bash -c "python tools/text_check.py > $OUT/text_check.txt 2>&1; tail -7 $OUT/text_check.txt"
step bench 400 bash -c "python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json"
[ "${2:-}" = quick ] && exit 0
step prof2 300 bash -c "rocprofv3 --kernel-trace --stats -d $OUT/prof2 -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra-configs > $OUT/prof2.log 2>&1; python tools/prof_stats.py $OUT/prof2 12"
step prof1 300 bash -c "MI_CLIP_PARTS=1 rocprofv3 --kernel-trace --stats -d $OUT/prof1 -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra-configs > $OUT/prof1.log 2>&1; python tools/prof_stats.py $OUT/prof1 12"
step pmcf 300 bash -c "rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $OUT/pmcf.log 2>&1"
step pmcw 300 bash -c "rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $OUT/pmcw.log 2>&1"
step pmc 60 bash -c "python tools/pmc_collect.py $OUT/pmc_f $OUT/pmc_w > $OUT/pmc.json 2>&1; cp profiles/pmc_latest.json $OUT/pmc_latest.json; head -20 $OUT/pmc.json"
step textprof 200 bash -c "rocprofv3 --kernel-trace --stats -d $OUT/proft -o t --output-format csv -- python3 tools/text_profile.py > $OUT/proft.log 2>&1; python tools/prof_stats.py $OUT/proft 12"
# what goes into profiles/ (cp gpurun_out/$R/commit/* profiles/), stamped HERE with the fingerprint of the sources that were measured
step commit 60 bash -c "mkdir -p $OUT/commit; \
  cp \$(ls $OUT/prof2/*/*kernel_stats.csv $OUT/prof2/*kernel_stats.csv 2>/dev/null | head -1) $OUT/commit/${R}_bench_kernel_stats.csv; \
  cp \$(ls $OUT/prof1/*/*kernel_stats.csv $OUT/prof1/*kernel_stats.csv 2>/dev/null | head -1) $OUT/commit/${R}_bench_kernel_stats_single_stream.csv; \
  cp \$(ls $OUT/proft/*/*kernel_stats.csv $OUT/proft/*kernel_stats.csv 2>/dev/null | head -1) $OUT/commit/${R}_text_query_kernel_stats.csv; \
  cp $OUT/pmc_latest.json $OUT/commit/pmc_latest.json; cp $OUT/bench.json $OUT/commit/${R}_bench.json; cp $OUT/text_check.txt $OUT/commit/${R}_text_check.txt; \
  for f in $OUT/commit/${R}_bench_kernel_stats.csv $OUT/commit/${R}_bench_kernel_stats_single_stream.csv $OUT/commit/pmc_latest.json; do python tools/profile_meta.py stamp \$f 256; done; ls $OUT/commit"
# keep the merged output small: the traces are large, the stats are what is kept
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete
exit 0
