"""per-round phase durations of conv3x3_patch_kernel (P3_TIMING library)"""
import os, subprocess, sys
HERE = "/root/repo/tools"
env = dict(os.environ, RYOLO_LIB=os.path.join(HERE, "variants", "lib_p3_timing.so"), P3_TIMING="1")
src = open(os.path.join(HERE, "bench_conv.py")).read().replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(os.path.dirname(HERE)))
src += '''
torch.cuda.synchronize()
d = dbg.view(-1, 4)
n = int((d[:, 3] > 0).sum())
d = d[:n].double().cpu()
t0 = d[:, 0].min()
for lo in range(0, n, 512):
    e = d[lo:lo + 512]
    print("blocks %5d..%5d  start %8.0f +- %6.0f  prologue %6.0f  loop %7.0f  epilogue %6.0f (min %6.0f max %6.0f)  end %8.0f" % (lo, lo + len(e) - 1,
          float((e[:, 0] - t0).mean()), float((e[:, 0] - t0).std()), float((e[:, 1] - e[:, 0]).mean()), float((e[:, 2] - e[:, 1]).mean()),
          float((e[:, 3] - e[:, 2]).mean()), float((e[:, 3] - e[:, 2]).min()), float((e[:, 3] - e[:, 2]).max()), float((e[:, 3] - t0).mean())))
'''
sys.exit(subprocess.run([sys.executable, "-c", src] + sys.argv[1:], env=env).returncode)
