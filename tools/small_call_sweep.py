import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hirest_amd
from hirest_amd import synth, ops
dev = torch.device('cuda:0')
model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval(); model.init_random_(seed=7); model.set_precision('bf16')
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randn((1024, 3, 224, 224), device=dev, generator=g).to(torch.bfloat16)
for mp in (False, None, False, None):
    ops.attention_set_mapping(mp)
    line = []
    for n in (64, 96, 128, 192, 256, 1024):
        x = frames[:n]; reps = max(3, 1024 // n)
        for _ in range(2): model.encode_image(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): model.encode_image(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        line.append(f"{n}: {n / dt:6.0f}")
    print("mapping", "per frame" if mp is False else "automatic", "  ".join(line), flush=True)
