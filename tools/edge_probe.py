import sys, torch
sys.path.insert(0, '/root/repo')
import hirest_amd
from hirest_amd import synth, retrieval
dev = torch.device('cuda:0')
m = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_TINY).to(dev).eval(); m.init_random_(seed=3)
for prec in ('bf16', 'fp32', 'bf16x3'):
    m.set_precision(prec)
    try:
        o = m.encode_image(torch.empty((0, 3, 224, 224), device=dev)); print(prec, 'empty images ->', tuple(o.shape), o.dtype)
    except Exception as e: print(prec, 'empty images: ERR', type(e).__name__, str(e)[:120])
    try:
        o = m.encode_text(torch.empty((0, 77), dtype=torch.long, device=dev)); print(prec, 'empty text ->', tuple(o.shape))
    except Exception as e: print(prec, 'empty text: ERR', type(e).__name__, str(e)[:120])
    o = m.encode_image(torch.randn((1, 3, 224, 224), device=dev)); print(prec, 'one image ->', tuple(o.shape), bool(torch.isfinite(o).all()))
    o = m.encode_image(torch.randn((65, 3, 224, 224), device=dev)); print(prec, '65 images ->', tuple(o.shape), bool(torch.isfinite(o).all()))
m.set_precision('bf16')
try:
    s = retrieval.score_corpus(torch.randn(0, 64, device=dev), torch.randn(5, 64, device=dev), [f'v{i}' for i in range(5)], [])
    print('no prompts ->', len(s), s.scores.shape)
except Exception as e: print('no prompts: ERR', type(e).__name__, str(e)[:160])
try:
    s = retrieval.score_corpus(torch.randn(3, 64, device=dev), torch.randn(1, 64, device=dev), ['v0'], ['a', 'b', 'c'])
    v, i = s.topk(10); print('one video, k=10 ->', v.shape, i.shape)
except Exception as e: print('one video: ERR', type(e).__name__, str(e)[:160])
