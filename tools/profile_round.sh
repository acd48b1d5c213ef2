#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_round.sh <tag>
# Pass 1: kernel trace + stats of the default bench command.  Passes 2/3: HBM traffic counters, one PMC pass each
# (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"), no other trace domain.
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/bench.py --no_cpu_baseline > $out/trace_bench.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 $root/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_graph --profile_steps 0 > $out/pmc_fetch.json 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 $root/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_graph --profile_steps 0 > $out/pmc_write.json 2> $out/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -o run -- python3 $root/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_graph --profile_steps 0 > $out/pmc_mfma.json 2> $out/pmc_mfma.log
ls -R $out | head -40
