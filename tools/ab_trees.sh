# ON THE GPU BOX: same-box A/B of two trees' default bench line, alternating:  bash tools/ab_trees.sh <other tree dir> [rounds] [bench flags ...]
other=$1; n=${2:-3}; shift 2
one() { ( cd $1 && python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_other_configs "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],4), d['lib_source_hash'])" ); }
for i in $(seq $n); do one . "$@"; one $other "$@"; done
