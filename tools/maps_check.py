import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, neural_lam_amd
torch.zeros(1, device="cuda")
libs = set()
for l in open("/proc/self/maps"):
    if "amdhip" in l or "hsa-runtime" in l or "nlam" in l:
        libs.add(l.split()[-1])
print("\n".join(sorted(libs)))
