cd $GRAFT_REPO_ROOT
for m in 1 3; do RYOLO_WGRAD_P1=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b8 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
print('P1=$m', d['value'], d['ms_per_step'], 'instr', d['kernel_timing']['ms_per_step_instrumented'], {n:(v['ms_per_step'],v['tflops']) for n,v in k.items() if 'wgrad' in n})"; done
for m in 1 3; do RYOLO_WGRAD_STREAM=0 RYOLO_WGRAD_P1=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b8 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('serial P1=$m', d['value'], d['ms_per_step'])"; done
