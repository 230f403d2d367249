cd /tmp
for lm in 64 8; do for B in 8 16 24 32 48 64 96; do
v=$(HUDIFF_LANE_MIN_B=$lm python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --pmc off --only-main 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'])")
echo "lane_min_b $lm B $B : $v"; done; done
