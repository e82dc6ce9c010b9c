import sys, numpy as np, torch
sys.path.insert(0, '.')
import oracle
from keypoint_bench_amd.utils.matcher import match_descriptors
g = np.load('tests/golden/match.npz')
for name in g['cases']:
    maxd, cc = g[name + '.prm']
    d0 = torch.from_numpy(g[name + '.sdesc0']).cuda(); d1 = torch.from_numpy(g[name + '.sdesc1']).cuda()
    pairs, dist = match_descriptors(d0, d1, max_distance=float(maxd), cross_check=bool(cc), return_distance=True)
    torch.cuda.synchronize()
    p = pairs.cpu().numpy(); d = dist.cpu().numpy()
    wp, wd = g[name + '.pairs'], g[name + '.dist']
    print(name, d0.shape, d1.shape, 'got', p.shape, 'want', wp.shape, 'pairs_eq', p.shape == wp.shape and (p == wp).all(),
          'dist_eq', d.shape == wd.shape and (d == wd).all())
    if p.shape == wp.shape and not (d == wd).all():
        bad = np.nonzero(d != wd)[0][:5]; print('  dist diffs', bad, d[bad], wd[bad], (d[bad]-wd[bad]))
    elif p.shape != wp.shape or not (p == wp).all():
        print('  got', p[:8].tolist(), 'want', wp[:8].tolist())
