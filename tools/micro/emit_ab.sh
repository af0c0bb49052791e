for e in 1 0; do
  ADVOC_EMIT_IMAGES=$e python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null > /tmp/b_$e.json
  python - <<PY
import json
r = json.load(open('/tmp/b_$e.json'))
print('emit', $e, 'ms_per_step', r['ms_per_step'])
PY
done
