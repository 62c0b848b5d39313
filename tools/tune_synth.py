import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
dev = torch.device('cuda', 0); ctx = ops.get_context(dev)
x = bench.synthetic_blocks(8, dev, 0)
print('input occupancy', float(x.mean()), 'points/block', float(x.sum() / 8))
model = ModelConfigType['c3p'].build(batch_size=8); model.compress([1, 1, 64, 64, 64])
base = model.get_weights()
for ga in (1.25, 1.35):
  for gs in (1.8, 2.0):
    for fb in (-1.0, -0.4, 0.0, 0.2):
        w = dict(base); rng = np.random.default_rng(43)
        for k in list(w):
            if k.endswith('/kernel'):
                g = ga if k.startswith(('analysis', 'hyper')) else gs
                w[k] = (w[k] * g).astype(np.float32)
            elif k.endswith('/bias') and not k.startswith('entropy'):
                w[k] = rng.normal(0, 0.05, w[k].shape).astype(np.float32)
        last = max(int(k.split('/')[1]) for k in w if k.startswith('synthesis/'))
        w[f'synthesis/{last}/bias'] = np.array([fb], np.float32)
        model.set_weights(w)
        enc = model._encode_batch(ctx, x, True)
        strings = enc['finish']()
        xyz, cnt = model._extract_points(ctx, enc['x_hat'], [128] * 8, True)
        d = enc['debug'][0]
        print(f'ga={ga} gs={gs} fb={fb}: ybytes={np.mean([len(s[0]) for s in strings]):8.0f} zbytes={np.mean([len(s[1]) for s in strings]):6.0f} '
              f'pts={float(cnt.float().mean()):9.0f} nz_sym={np.mean(d["symbols"]!=0):.3f} |sym|max={np.abs(d["symbols"]).max()} sigma_mean={d["sigma_hat"].mean():.3f} xhat_max={d["x_hat"].max():.3f}')
