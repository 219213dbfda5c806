"""Register / scratch / occupancy report of the kernels of one .hip file (device-only compile with -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_regs.py vae_segmentation_amd/csrc/igemm_k3_bf16.hip [name filter] [-DVS_DET_BUILD=1 ...]"""
import re, subprocess, sys, tempfile, os

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-variable", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "dev.o")] + extra, capture_output=True, text=True)
keys = ("VGPRs:", "AGPRs:", "ScratchSize [bytes/lane]:", "VGPRs Spill:", "Occupancy [waves/SIMD]:", "LDS Size [bytes/block]:")
cur, rec = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)|Name: (\S+)", line)
    if m:
        cur, rec = (m.group(1) or m.group(2)), {}
    for k in keys:
        if k in line and cur:
            rec[k] = line.split(k)[1].split("[")[0].strip()
    if "LDS Size" in line and cur:
        name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip() or cur
        if filt in name:
            print("%-96s vgpr %s agpr %s scratch %s spill %s occ %s" % (name[:96], rec.get(keys[0]), rec.get(keys[1]), rec.get(keys[2]), rec.get(keys[3]), rec.get(keys[4])))
        cur = None
if r.returncode:
    print(r.stderr[-2000:])
    sys.exit(r.returncode)
