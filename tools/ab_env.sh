# Tuning aid: same-box A/B of the whole bench line for an environment switch (usage: bash tools/ab_env.sh VAR=value), alternating twice.
for i in 1 2; do
  for v in off on; do
    if [ $v = off ]; then env "$1" python bench.py --no-cpu-baseline --no-ratio320 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], (d.get('other_entry') or {}).get('ms_per_step'))"
    else python bench.py --no-cpu-baseline --no-ratio320 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], (d.get('other_entry') or {}).get('ms_per_step'))"
    fi
  done
done
