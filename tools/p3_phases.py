"""Cycle counters per phase of conv3x3_patch_kernel (library built with -DP3_TIMING, RYOLO_LIB=tools/variants/lib_p3_timing.so):
prologue (-> first MFMA step), main loop, epilogue; averaged over the workgroups.  Args as tools/bench_conv.py."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
env = dict(os.environ, RYOLO_LIB=os.path.join(HERE, "variants", "lib_p3_timing.so"), P3_TIMING="1")
src = open(os.path.join(HERE, "bench_conv.py")).read().replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(os.path.dirname(HERE)))
src += '''
torch.cuda.synchronize()
d = dbg.view(-1, 4)[:35000]
d = d[d[:, 3] > 0].double()
print("workgroups", d.shape[0], "cycles/WG: prologue %.0f  loop %.0f  epilogue %.0f  total %.0f" % (
    float((d[:, 1] - d[:, 0]).mean()), float((d[:, 2] - d[:, 1]).mean()), float((d[:, 3] - d[:, 2]).mean()), float((d[:, 3] - d[:, 0]).mean())))
e = dbg.view(-1, 4)[35000:35000 + d.shape[0]].double()
print("epilogue of wave 0: barrier %.0f  stage writes %.0f  statistics %.0f  store loops %.0f" % tuple(float(e[:, k].mean()) for k in range(4)))
'''
sys.exit(subprocess.run([sys.executable, "-c", src] + sys.argv[1:], env=env).returncode)
