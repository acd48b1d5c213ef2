for i in 1 2 3; do for v in "512 1024" "128 1024" "512 256" "64 1024"; do set -- $v
T3D_FIN_BIG=$1 T3D_FIN_WIDE=$2 python bench.py --steps 200 --warmup 30 --no_other_configs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); k=d['roofline']['per_kernel_us_per_step']; print('FIN_BIG=$1 FIN_WIDE=$2', d['ms_per_step'], {n: round(v,1) for n, v in k.items() if 'finalize' in n})"
done; done
