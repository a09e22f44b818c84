cd /root/repo
python bench.py --mbytes 1024 --steps 5 --warmup 2 --hot-path-only --no-cpu-baseline 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ms', j['ms_per_step'], j['roofline']['kernel_ms'], 'verified', j['config']['verified_docs_vs_oracle'])
    elif 'INVALID' in l or 'rror' in l: print(l)
"
