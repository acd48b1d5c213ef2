root=$(pwd); out=$root/gpurun_out/pmc_lds; mkdir -p $out; cd /tmp && export TMPDIR=/tmp
small="--dtype bf16 --batch_size 128 --num_point 2048 --steps 3 --warmup 2 --no_cpu_baseline --no_other_configs --no_graph --profile_steps 0"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/a -o run -- python3 $root/bench.py $small > $out/a.json 2> $out/a.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/b -o run -- python3 $root/bench.py $small > $out/b.json 2> $out/b.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $out/c -o run -- python3 $root/bench.py $small > $out/c.json 2> $out/c.log
cd $root
for x in a b c; do echo "== pass $x"; python3 tools/pmc_generic.py $out/$x --per_cycle GRBM_GUI_ACTIVE 2>&1 | head -14; tail -3 $out/$x.log | grep -i "error\|invalid" ; done
rm -rf $out/a $out/b $out/c
