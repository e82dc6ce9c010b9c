import os, sys, ctypes, numpy as np, torch
os.environ["KPB_NMS_DBG"] = "1"
sys.path.insert(0, '.')
from keypoint_bench_amd._lib import Context
from keypoint_bench_amd.utils.extracter import detection_batch
g = np.load('tests/golden/alike_t.npz')
B = 128
s = torch.from_numpy(np.stack([g['full.score0'], g['full.score1']] * (B // 2)))[:, None].cuda().contiguous()
p = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
ctx = Context.get(s.device)
detection_batch(s, p, sync=False); ctx.sync()
ptr, n = open('/tmp/kpb_nms_dbg_ptr').read().split()
n = int(n)
hip = ctypes.CDLL([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0])
buf = np.zeros(n, np.int64)
hip.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(int(ptr, 16)), ctypes.c_size_t(n * 8), 2)
d = buf.reshape(16, B, 150, 8)
for sw in range(6):
    x = d[sw][d[sw][..., 6] == 1]
    if len(x) == 0: print(sw, 'no tiles'); continue
    print('sweep %d tiles %5d  load %6.0f  H %6.0f  V %6.0f  K %6.0f  iters %.2f  total %7.0f cycles' % (sw, len(x), x[:, 0].mean(), x[:, 1].mean(), x[:, 2].mean(), x[:, 3].mean(), x[:, 4].mean(), x[:, 5].mean()))
