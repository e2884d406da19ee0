#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box from the repo root; outputs under gpurun_out/refresh).
# usage: bash tools/refresh_profiles.sh <tag> <commit>     e.g. r02 abc1234
# Order: PMC passes first (their per-kernel traffic file is what bench.py reports as roofline.traffic), then the bench lines,
# then the kernel-trace statistics of the same command line.  PMC and --stats runs use --batch 1 --in-flight 1 so that the per-launch
# averages are those of single-blob launches with nothing else on the chip (what the bench line's instrumented replay times).
# usage (only the profiler passes): ONLY_PROF=1 bash tools/refresh_profiles.sh r02 <commit>
set -u
TAG=${1:-r02}
COMMIT=${2:-}
OUT=gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
ONE="--no-cpu-baseline --no-by-config --no-end-to-end --no-reconstruct --batch 1 --in-flight 1 --batch-extra 0 --sequential-extra 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 $ONE > /dev/null 2> $OUT/pmc_fetch_err.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 $ONE > /dev/null 2> $OUT/pmc_write_err.txt
# the same two passes in the measured loop's mode with 4 blobs per call, 2 calls in flight: launches cover 4 blobs each
BAT="--only-measured-loop --batch 4"  # (4 blobs per launch: what tools/traffic_from_pmc.py divides the batched figures by; the default policy gives 5 at 2^24)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b -- python3 bench.py --steps 8 --warmup 0 $BAT > /dev/null 2> $OUT/pmc_fetch_b_err.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b -- python3 bench.py --steps 8 --warmup 0 $BAT > /dev/null 2> $OUT/pmc_write_b_err.txt
python tools/traffic_from_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_prove24_traffic.json "$COMMIT" $OUT/pmc_fetch_b $OUT/pmc_write_b
cp $OUT/${TAG}_prove24_traffic.json profiles/${TAG}_prove24_traffic.json
cp $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $OUT/${TAG}_prove24_pmc_FETCH_SIZE.csv
cp $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/${TAG}_prove24_pmc_WRITE_SIZE.csv
cp $(ls $OUT/pmc_fetch_b/*/*counter_collection.csv | head -1) $OUT/${TAG}_prove24_batched_pmc_FETCH_SIZE.csv
cp $(ls $OUT/pmc_write_b/*/*counter_collection.csv | head -1) $OUT/${TAG}_prove24_batched_pmc_WRITE_SIZE.csv
if [ -z "${ONLY_PROF:-}" ]; then
python bench.py > $OUT/${TAG}_prove24_bench.json 2> $OUT/prove_err.txt
python bench.py --workload commit > $OUT/${TAG}_commit24_bench.json 2> $OUT/commit_err.txt
python bench.py --log-domain 22 --cpu-sample-log 22 > $OUT/${TAG}_prove22_bench.json 2> $OUT/prove22_err.txt
python bench.py --log-domain 20 --cpu-sample-log 20 > $OUT/${TAG}_prove20_bench.json 2> $OUT/prove20_err.txt
python bench.py --workload commit --log-domain 22 --cpu-sample-log 22 > $OUT/${TAG}_commit22_bench.json 2> $OUT/commit22_err.txt
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 $ONE > $OUT/stats_run.json 2> $OUT/stats_err.txt
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_prove24_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats22 -- python3 bench.py --log-domain 22 --steps 10 --warmup 3 $ONE > $OUT/stats22_run.json 2> $OUT/stats22_err.txt
cp $(ls $OUT/stats22/*/*kernel_stats.csv | head -1) $OUT/${TAG}_prove22_kernel_stats.csv
# the measured loop's own mode (4 blobs per call, 2 calls in flight): what the narrow launches cost while another batch's wide ones run
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b -- python3 bench.py --only-measured-loop --steps 24 --warmup 0 > /dev/null 2> $OUT/stats_b_err.txt
cp $(ls $OUT/stats_b/*/*kernel_stats.csv | head -1) $OUT/${TAG}_prove24_batched_kernel_stats.csv
# VALU utilisation of the wide kernels: SQ counters (their own pass: counters + kernel trace only), tools/valu_util.py
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 4 --warmup 1 $ONE > /dev/null 2> $OUT/pmc_sq_err.txt
python tools/valu_util.py $OUT/pmc_sq > $OUT/${TAG}_valu_util_prove24.txt 2>> $OUT/pmc_sq_err.txt
rm -rf $OUT/stats $OUT/stats22 $OUT/stats_b $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_b $OUT/pmc_write_b
bash tools/build_evidence.sh $OUT/${TAG}_build_evidence.txt  # clean-build wall time, offload bundles by target, kernel count (no GPU needed)
ls -la $OUT
