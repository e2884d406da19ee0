#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box from the repo root; outputs under gpurun_out/refresh).
# usage: bash tools/refresh_profiles.sh <tag>      e.g. r01_final
set -u
TAG=${1:-r01_final}
OUT=gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_prove24_bench.json 2> $OUT/prove_err.txt
python bench.py --workload commit > $OUT/${TAG}_commit24_bench.json 2> $OUT/commit_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --pipeline-depth 0 > $OUT/stats_run.json 2> $OUT/stats_err.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipeline-depth 0 > /dev/null 2> $OUT/pmc_fetch_err.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipeline-depth 0 > /dev/null 2> $OUT/pmc_write_err.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_prove24_kernel_stats.csv
python tools/traffic_from_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/r01_prove24_traffic.json
cp $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $OUT/r01_prove24_pmc_FETCH_SIZE.csv
cp $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/r01_prove24_pmc_WRITE_SIZE.csv
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
