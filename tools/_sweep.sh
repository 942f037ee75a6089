for cfg in "48 2" "32 2" "40 2" "56 2" "80 2"; do
  set -- $cfg
  python bench.py --steps 12 --warmup 2 --batch $1 --sub-batches $2 --cpu-frames 0 --no-kernel-events --no-ingest --no-parity 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value']), d['step_ms']['median'])"
done
