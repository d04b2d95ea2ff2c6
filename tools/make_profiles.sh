#!/bin/bash
# Regenerates the per-round measurement artefacts on the GPU box (run through gpurun from the repository root):
#   bash tools/make_profiles.sh r02        -> gpurun_out/r02_*  (copy what is to be judged into profiles/)
set -o pipefail
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
$B --steps 20 --warmup 5 > $OUT/${R}_bench_n1.json 2> $OUT/${R}_bench_n1.err
$B --workload basic --steps 10 --warmup 3 > $OUT/${R}_bench_basic.json 2> $OUT/${R}_bench_basic.err
$B --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${R}_bench_512_bf16.json 2> $OUT/${R}_bench_512_bf16.err
$B --fp8 --size 512 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${R}_bench_512_fp8.json 2> $OUT/${R}_bench_512_fp8.err
echo "[profiles] bench lines done"
# per-kernel durations, one stream (not stretched by concurrent kernels) and the default three streams
GAN_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${R}_prof_single -o ks -- $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${R}_prof_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${R}_prof_multi -o ks -- $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${R}_prof_multi.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${R}_prof_fp8 -o ks -- $B --fp8 --size 512 --batch 8 --steps 8 --warmup 3 --no-cpu-baseline > $OUT/${R}_prof_fp8.log 2>&1
echo "[profiles] kernel traces done"
# HBM traffic: separate PMC passes, no trace domains
GAN_SINGLE_STREAM=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${R}_pmc_fetch -o pf -- $B --steps 2 --warmup 2 --no-cpu-baseline > $OUT/${R}_pmc_fetch.log 2>&1
GAN_SINGLE_STREAM=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${R}_pmc_write -o pw -- $B --steps 2 --warmup 2 --no-cpu-baseline > $OUT/${R}_pmc_write.log 2>&1
echo "[profiles] pmc passes done"
cd $ROOT
python3 tools/prof_summary.py $OUT/${R}_prof_single 25 > $OUT/${R}_kernel_stats_summary.txt
cp $(ls $OUT/${R}_prof_single/*kernel_stats.csv | head -1) $OUT/${R}_kernel_stats.csv
python3 tools/prof_summary.py $OUT/${R}_prof_multi 25 > $OUT/${R}_kernel_stats_three_streams_summary.txt
python3 tools/prof_summary.py $OUT/${R}_prof_fp8 11 > $OUT/${R}_kernel_stats_512_fp8_summary.txt
python3 tools/timeline.py $OUT/${R}_prof_multi > $OUT/${R}_timeline.txt
python3 tools/step_trace.py $OUT/${R}_prof_multi 2 > $OUT/${R}_step_trace.txt
python3 tools/pmc_traffic.py $OUT/${R}_pmc_fetch $OUT/${R}_pmc_write $OUT/${R}_pmc_traffic.json 2> $OUT/${R}_pmc_traffic.txt
python3 tools/step_ops.py > $OUT/${R}_step_ops.txt 2> $OUT/${R}_step_ops.err
python3 tools/bench_norm.py 32 64 64 256 > $OUT/${R}_norm_kernels.txt 2>/dev/null
python3 tools/bench_norm.py 16 64 64 256 >> $OUT/${R}_norm_kernels.txt 2>/dev/null
python3 tools/bench_fp8.py 32 64 > $OUT/${R}_fp8_kernels.txt 2>/dev/null
python3 tools/bench_fp8.py 8 128 >> $OUT/${R}_fp8_kernels.txt 2>/dev/null
rm -rf $OUT/${R}_prof_single $OUT/${R}_prof_multi $OUT/${R}_prof_fp8 $OUT/${R}_pmc_fetch $OUT/${R}_pmc_write
echo "[profiles] summaries written"
