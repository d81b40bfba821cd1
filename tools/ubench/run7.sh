cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_run10; mkdir -p $o
for v in 1 0 1 0; do
  TOK_GEMM256=$v python bench.py --backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('swin TOK_GEMM256=$v', d['ms_per_step'])"
done | tee $o/ab.txt
for v in 1 0; do
  TOK_GEMM256=$v python bench.py --backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('hrnet TOK_GEMM256=$v', d['ms_per_step'])"
  TOK_GEMM256=$v python bench.py --backbone davit_t --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('davit TOK_GEMM256=$v', d['ms_per_step'])"
done | tee -a $o/ab.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -k "gemm256" -x -q 2>&1 | tail -2
