# python bench.py value against the number of batches in flight (same box, same call): bash tests/manual/inflight_sweep.sh [n ...]
for n in ${@:-8 12 16 24 32}; do
python bench.py --in-flight $n --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('in flight', $n, 'value', round(d['value']), 'long run', round(d.get('value_long_run') or 0))"
done
