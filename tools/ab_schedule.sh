cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for S in 1 2; do
python bench.py --no-cpu-baseline --steps 400 --schedule $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('schedule $S round $r: %.3f M env-steps/s kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_avg_ms']))"
done; done
python tools/phase_profile.py 4096
