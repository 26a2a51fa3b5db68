"""Dev tool (GPU box): per-kernel profile of the inference forward at batch 128."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oct_image_segmentation_models_amd.engine import UNetEngine
from oct_image_segmentation_models_amd.common.synthetic import make_scans
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = UNetEngine(device="cuda:0", input_channels=1, num_classes=3, image_height=256, image_width=512, max_batch=B)
im, _ = make_scans(8, 256, 512, 3)
x = torch.from_numpy(np.tile(im, (B // 8, 1, 1, 1))).cuda()
for _ in range(2): eng.forward(x, want_argmax=True)
eng.profile_begin()
for _ in range(3): eng.forward(x, want_argmax=True)
ents = eng.profile_end()
tot = sum(e["total_ms"] for e in ents) / 3
print("B", B, "total kernel ms", round(tot, 3), "=> ms/scan", round(tot / B, 5))
for e in sorted(ents, key=lambda e: -e["total_ms"]):
    ms = e["total_ms"] / e["launches"]
    print(f"{e['layer']:12s} {e['kernel']:36s} {ms:7.3f} {e['flops']/e['launches']/ms/1e9:6.1f} TF {e['bytes']/e['launches']/ms/1e6:7.0f} GB/s")
