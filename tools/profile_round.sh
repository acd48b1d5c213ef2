#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_round.sh <tag> [bench.py flags of the workload ...]
#   bash tools/profile_round.sh r02_f32
#   bash tools/profile_round.sh r02_bf16 --dtype bf16 --batch_size 128 --num_point 2048
# Pass 1: kernel trace + stats of the bench command.  Passes 2/3: HBM traffic counters, one PMC pass each
# (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"), no other trace domain.
# Pass 4: MFMA-busy cycles + clock.
tag=${1:-r03}
[ $# -gt 0 ] && shift
flags="$@"
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py $flags > $out/bench.json 2> $out/bench.err
# (--no_other_configs in every profiler pass: the side runs of configs[2..4] are child processes that inherit the profiler and would
# write their own counter / stats files into the same directories)
# --profile_steps 0: no per-launch timing leg (bench.py's un-hosted eager launches of every kernel, which round 5's stats mixed with the
# hosted graph replays: 994 calls of the dominant kernel, min 10.6 / max 58.3 us).  What the stats file then holds is one eager step and
# W + K hipGraph replays of the step: its per-kernel averages ARE the hosted durations, and `roofline.avg_launch_us` of a plain bench
# run (events around the hosted replays of the dominant family) can be checked against it without the bench line.
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/bench.py $flags --no_cpu_baseline --no_other_configs --profile_steps 0 > $out/trace_bench.json 2> $out/trace.log
small="--steps 4 --warmup 2 --no_cpu_baseline --no_other_configs --no_graph --profile_steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 $root/bench.py $flags $small > $out/pmc_fetch.json 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 $root/bench.py $flags $small > $out/pmc_write.json 2> $out/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -o run -- python3 $root/bench.py $flags $small > $out/pmc_mfma.json 2> $out/pmc_mfma.log
cd $root
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write -o $out/pmc_traffic.json $(echo "$flags" | sed 's/--steps [0-9]*//; s/--warmup [0-9]*//') > $out/pmc_traffic.txt 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma -o $out/pmc_mfma.json > $out/pmc_mfma.txt 2>&1
# the one process that was profiled wrote one stats file; more than one means a child slipped in: take the largest and say so
nstats=$(find $out/trace -name "*kernel_stats.csv" | wc -l)
[ "$nstats" != "1" ] && echo "WARNING: $nstats kernel_stats.csv files under $out/trace" | tee -a $out/trace.log
cp "$(find $out/trace -name "*kernel_stats.csv" -printf '%s %p\n' | sort -nr | head -1 | cut -d' ' -f2-)" $out/kernel_stats.csv 2>/dev/null
# raw traces are large; they go only once every summary exists
[ -s $out/pmc_traffic.json ] && [ -s $out/pmc_mfma.json ] && [ -s $out/kernel_stats.csv ] && rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_mfma
ls -la $out
