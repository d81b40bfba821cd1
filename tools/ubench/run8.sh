cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_run11; mkdir -p $o
for v in 1 2 1 2; do
  TOK_GEMM256=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('resnet50 TOK_GEMM256=$v', d['ms_per_step'])"
done | tee $o/ab.txt
for v in 1 2 1 2; do
  TOK_GEMM256=$v python bench.py --backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('swin TOK_GEMM256=$v', d['ms_per_step'])"
done | tee -a $o/ab.txt
for k in 512 1024; do
  TOK_GEMM256=2 TOK_GEMM256_MIN_K=$k python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('resnet50 TOK_GEMM256=2 MIN_K=$k', d['ms_per_step'])"
done | tee -a $o/ab.txt
