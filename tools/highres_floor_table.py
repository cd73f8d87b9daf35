"""The launches of the first five layers (800^2 ... 200^2 maps, <= 128 channels) of the batch-64 step against their BYTE floors
(VERDICT r4 item 3): floor = algorithmic bytes (every operand element once) / 6.3 TB/s (the copy rate this part reaches).  Reads a per-launch
table of tools/profile_layers.py (B=64, streams serialized: every launch timed alone; the CU-exclusive 8-wave weight gradient therefore on
the 96 CUs it is sized for: its row also gives the chip time, us x 96 / 256) and prints one row per launch, with the matrix floor (FLOPs / 2.5 PFLOP/s)
beside the byte floor: the 64-channel 3x3 layers are bound by the larger of the two.  usage: python tools/highres_floor_table.py gpurun_out/r05m/per_launch_b64.txt"""
import re
import sys

RATE = 6.3e12
B = 64
rows = []
stem_bytes = {  # csrc/stem.hip: statistics pass reads the fp32 image; forward reads it and writes the bf16 output; backward reads dz + image
    "stats": B * 3 * 800 * 800 * 4, "fwd": B * 3 * 800 * 800 * 4 + B * 800 * 800 * 32 * 2, "bwd": B * 800 * 800 * 32 * 2 + B * 3 * 800 * 800 * 4}
stem_seen = {"fwd": 0}
for ln in open(sys.argv[1]):
    m = re.match(r"(fwd|bwd) #\s*(\d+) (\S+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) GB/s\s+flops\s+([\d.]+) G\s+bytes\s+([\d.]+) MB\s*(.*)", ln)
    if not m:
        continue
    d, idx, kern, us, tf, gbs, gf, mb, desc = m.groups()
    us, mb = float(us), float(mb)
    if kern.startswith("ryolo_stem3x3"):
        if d == "bwd":
            which = "bwd"
        else:
            which = "stats" if us < 600 else "fwd"
        rows.append((us, f"{d} {kern} ({which})", stem_bytes[which] / 1e6, "3->32 3x3 800x800", 2.0 * B * 800 * 800 * 32 * 27 / 1e9 * (2 if which == "bwd" else 1)))
    elif re.search(r"(400x400|800x800)", desc) or re.search(r"128->64 taps1x4 200x200", desc) or re.search(r"64->128 taps9 200x200", desc):
        rows.append((us, f"{d} {kern}", mb, desc, float(gf)))
print(f"{'launch':52s} {'shape':34s} {'us':>8s} {'MB':>8s} {'byte fl.':>9s} {'x':>6s} {'GFLOP':>8s} {'mfma fl.':>9s} {'x larger':>9s}")
tot = fl = fl2 = 0.0
for us, name, mb, desc, gf in sorted(rows, reverse=True):
    note = ""
    if "conv3x3_wgrad_kernel" in name:                      # the 8-wave ring kernel on 96 of 256 CUs (engine/graph.py reports its class under this name)
        note = f"   <- CU-exclusive on 96 CUs: chip time {us * 96 / 256:.1f} us"
        us_eff = us * 96 / 256
    else:
        us_eff = us
    floor = mb * 1e6 / RATE * 1e6
    mf = gf * 1e9 / 2.5e15 * 1e6
    tot += us_eff
    fl += floor
    fl2 += max(floor, mf)
    print(f"{name:52s} {desc:34s} {us:8.1f} {mb:8.1f} {floor:9.1f} {us_eff / floor:6.2f} {gf:8.1f} {mf:9.1f} {us_eff / max(floor, mf):9.2f}{note}")
print(f"{'sum (chip time for the CU-exclusive row)':87s} {tot:8.1f} {'':8s} {fl:9.1f} {tot / fl:6.2f} {'':8s} {fl2:9.1f} {tot / fl2:9.2f}")
