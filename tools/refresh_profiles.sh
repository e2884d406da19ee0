#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box from the repo root; outputs under gpurun_out/refresh).
# usage: bash tools/refresh_profiles.sh <tag>      e.g. r01_final
# Order: PMC passes first (their per-kernel traffic file is what bench.py reports as roofline.traffic), then the bench lines,
# then the kernel-trace statistics of the same command line.
set -u
TAG=${1:-r01_final}
OUT=gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
ONE="--no-cpu-baseline --pipeline-depth 0 --batch-extra 0"   # one proof at a time only: the averages are per single-blob launch
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 $ONE > /dev/null 2> $OUT/pmc_fetch_err.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 $ONE > /dev/null 2> $OUT/pmc_write_err.txt
python tools/traffic_from_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/r01_prove24_traffic.json
cp $OUT/r01_prove24_traffic.json profiles/r01_prove24_traffic.json
cp $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $OUT/r01_prove24_pmc_FETCH_SIZE.csv
cp $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/r01_prove24_pmc_WRITE_SIZE.csv
python bench.py > $OUT/${TAG}_prove24_bench.json 2> $OUT/prove_err.txt
python bench.py --workload commit > $OUT/${TAG}_commit24_bench.json 2> $OUT/commit_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 $ONE > $OUT/stats_run.json 2> $OUT/stats_err.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_prove24_kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
