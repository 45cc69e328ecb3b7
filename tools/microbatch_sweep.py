import sys, time, torch
sys.path.insert(0, '/root/repo')
import hirest_amd
from hirest_amd import synth
dev = torch.device('cuda:0')
model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
model.init_random_(seed=7)
model.set_precision('bf16')
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randn((2048, 3, 224, 224), device=dev, generator=g).to(torch.bfloat16)
for mb in (1024, 512, 256, 128, 96, 64):
    model.visual.max_frames_per_call = mb
    model.encode_image(frames[:mb * 2]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): model.encode_image(frames)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"frames per tower call {mb:5d}: {2048 / dt:8.1f} frames/s", flush=True)
