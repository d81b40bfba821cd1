# same-box A/B of two library builds on the headline workload (tools/ab_lib.sh <rev> builds libtok_ab.so)
run() { env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for i in 1 2 3; do
run TOK_LIB=torchok_amd/lib/libtok_ab.so
run A=new
done
