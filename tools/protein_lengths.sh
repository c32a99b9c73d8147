for rep in 1 2; do
python bench.py --config c4 --seg-len 200 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 200 aa: ms/step %.2f align %.2f  %.1f M/s' % (d['ms_per_step'], d['align_ms_per_step'], d['value']/1e6))"
python bench.py --config c4 --seg-len 130 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4 130 aa: ms/step %.2f align %.2f  %.1f M/s' % (d['ms_per_step'], d['align_ms_per_step'], d['value']/1e6))"
python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4  67 aa: ms/step %.2f align %.2f  %.1f M/s' % (d['ms_per_step'], d['align_ms_per_step'], d['value']/1e6))"
done
