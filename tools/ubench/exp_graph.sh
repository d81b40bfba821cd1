cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; "$@" 2>/tmp/err.txt | python -c "import json,sys; s=sys.stdin.read().strip(); d=json.loads(s) if s else {}; print('$tag', d.get('ms_per_step'), d.get('config',{}).get('launch_mode'))" || tail -3 /tmp/err.txt; }
H="python bench.py --backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 8 --warmup 3 --no-cpu-baseline"
run hr_eager $H --graph 0
run hr_graph $H --graph 1
TOK_SIDE_IN_GRAPH=1 run hr_graph_side $H --graph 1
S="python bench.py --backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline"
run sw_eager $S --graph 0
run sw_graph $S --graph 1
TOK_SIDE_IN_GRAPH=1 run sw_graph_side $S --graph 1
R="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run rn_graph $R --graph 1
TOK_SIDE_IN_GRAPH=1 run rn_graph_side $R --graph 1
