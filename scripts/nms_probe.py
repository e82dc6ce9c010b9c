import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from keypoint_bench_amd._lib import Context
from keypoint_bench_amd.utils.extracter import detection_batch
g = np.load('tests/golden/alike_t.npz')
s = torch.from_numpy(np.stack([g['full.score0'], g['full.score1']] * 64))[:, None].cuda().contiguous()
p = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
ctx = Context.get(s.device)
for _ in range(2): detection_batch(s, p, sync=False); ctx.sync()
ctx.prof_enable(True)
for _ in range(3): detection_batch(s, p, sync=False); ctx.sync()
r = ctx.prof_report()
print({k: (v[0], round(v[1] / 3, 3)) for k, v in r.items()})
