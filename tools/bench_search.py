import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops, model_opt
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
dev = torch.device('cuda', 0); ctx = ops.get_context(dev)
model = ModelConfigType['c3p'].build(batch_size=32); model.compress([1, 1, 64, 64, 64])
model.set_weights(bench.synthetic_weights(model))
x = bench.synthetic_blocks(32, dev, 0)
blocks = [np.argwhere(b.cpu().numpy() > 0).astype(np.float64) for b in x]
enc = model._encode_batch(ctx, x, False); enc['finish'](); x_hat = enc['x_hat']
thr = model.thresholds
model_opt.compute_optimal_thresholds_gpu(ctx, blocks, x_hat, thr, 64); torch.cuda.synchronize()
t0 = time.perf_counter(); names, best = model_opt.compute_optimal_thresholds_gpu(ctx, blocks, x_hat, thr, 64); torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
xh = np.clip(x_hat.cpu().numpy(), 0, 1)
t0 = time.perf_counter(); hb = [model_opt.compute_optimal_thresholds(blocks[i], xh[i], thr, 64, opt_metrics=['d1_mse'], max_deltas=[np.inf])[1] for i in range(4)]; t_host = (time.perf_counter() - t0) / 4
print(f'GPU adaptive search: {1e3*t_gpu:.1f} ms for 32 blocks ({1e3*t_gpu/32:.2f} ms/block); host KD-tree path: {1e3*t_host:.0f} ms/block -> {t_host/(t_gpu/32):.0f}x; decisions equal: {hb == best[:4]}; best idx sample {best[:6]}')

# D2 (normals): host KD-tree path, one block per worker process (model_opt.HostSearchPool) vs one after the other
nb = [np.hstack([b, np.random.default_rng(i).normal(size=b.shape)]) for i, b in enumerate(blocks)]
jobs = [(nb[i], xh[i], thr, 64, True, ['d1_mse', 'd2_mse'], [np.inf]) for i in range(32)]
t0 = time.perf_counter(); serial = [model_opt.compute_optimal_thresholds(nb[i], xh[i], thr, 64, normals=nb[i][:, 3:6], opt_metrics=['d1_mse', 'd2_mse'], max_deltas=[np.inf])[1] for i in range(2)]; t_ser = (time.perf_counter() - t0) / 2
pool = model_opt.HostSearchPool(32)
t0 = time.perf_counter(); par = pool.map(jobs); t_par = (time.perf_counter() - t0) / 32
pool.close()
print(f'D2 host search: serial {1e3*t_ser:.0f} ms/block; 32 worker processes {1e3*t_par:.0f} ms/block ({t_ser/t_par:.1f}x); decisions equal: {[list(map(int, s)) for s in serial] == [p[1] for p in par[:2]]}; cores {os.cpu_count()}')
