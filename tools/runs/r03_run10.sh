cd $GRAFT_REPO_ROOT
python -c "
import torch
print('prio range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p, s.priority)
    except Exception as e: print(p, 'ERR', e)
"
for pr in 0 1 -1 0 1; do RYOLO_SIDE_PRIO=$pr python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PRIO=$pr', d['value'], d['ms_per_step'], 'b8', d['b8']['value'], d['b8']['ms_per_step'])"; done
