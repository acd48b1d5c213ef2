# ON THE GPU BOX: same-box A/B of the config-4 (bf16 B=128 N=2048) step under environment switches / bench flags:
#   bash tools/il_ab.sh "A=1" "B=2 C=3 --no_graph" ...      (words starting with -- go to bench.py)
for i in 1 2; do
for spec in "$@"; do
( fl=""; for e in $spec; do case $e in --*) fl="$fl $e";; *) export $e;; esac; done
python bench.py --dtype bf16 --batch_size 128 --num_point 2048 --no_cpu_baseline --no_other_configs --steps 60 --warmup 10 $fl 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-60s' % '$spec', round(d['value']), round(d['ms_per_step'], 4))" )
done; done
