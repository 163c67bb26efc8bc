for k in 1 3; do for v in light full; do
python bench.py --workload config2 --variant $v --views-in-flight $k --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json
python -c "
import json; d=json.load(open('/tmp/l.json')); print('$v K=$k', round(d['ms_per_step'],4), 'ms/view', round(d['config']['views_per_s']), 'views/s')"
done; done
