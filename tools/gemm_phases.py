"""Cycle counters per phase of conv_gemm_kernel (library built with -DGEMM_TIMING, RYOLO_LIB=tools/variants/lib_gemm_timing.so):
prologue -> first MFMA, main loop, store, statistics; averaged over the workgroups.  Args as tools/bench_conv.py."""
import os, subprocess, sys
env = dict(os.environ, RYOLO_LIB=os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "lib_gemm_timing.so"), GEMM_TIMING="1")
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "bench_conv.py")).read().replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(os.path.dirname(HERE)))
src += '''
torch.cuda.synchronize()
M = B * OH * OH
nwg = ((M + 127) // 128) * ((Cout + 127) // 128)
d = dbg[:4 * min(nwg, 70000)].view(-1, 4).double()
print("workgroups", nwg, "cycles/WG: prologue %.0f  loop %.0f  store %.0f  tail %.0f  total %.0f" % tuple(list(d.mean(0).tolist()) + [float(d.sum(1).mean())]))
'''
sys.exit(subprocess.run([sys.executable, "-c", src] + sys.argv[1:], env=env).returncode)
