set -e
for a in "--steps 20 --warmup 5" "" "--envs 1024" "--envs 2048" "--envs 16384 --steps 300"; do
  timeout -k 10 200 python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a', '| %.3f M env-steps/s, ms/step %.4f, kernel %.4f pre %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['pre_kernel_avg_ms']))"
done
