"""Worker process of model_opt.HostSearchPool: reads pickled jobs (length-prefixed) from stdin, runs the reference's KD-tree
threshold search (model_opt.compute_optimal_thresholds == /root/reference/src/model_opt.py:21-77) and writes the pickled result
to stdout.  Plain numpy / scipy: no GPU, no torch."""
import pickle
import struct
import sys


def main():
    import numpy as np
    from pcc_geo_cnn_v2_amd.model_opt import compute_optimal_thresholds
    rd, wr = sys.stdin.buffer, sys.stdout.buffer
    while True:
        hdr = rd.read(8)
        if len(hdr) < 8:
            return
        job = pickle.loads(rd.read(struct.unpack('<Q', hdr)[0]))
        try:
            block, x_hat, thresholds, resolution, with_normals, opt_metrics, max_deltas = job
            normals = block[:, block.shape[1] - 3:] if with_normals else None
            names, best = compute_optimal_thresholds(block, x_hat, thresholds, resolution, normals=normals, opt_metrics=opt_metrics,
                                                     max_deltas=max_deltas, fixed_threshold=False)
            out = ('ok', names, [int(b) for b in best])
        except BaseException as e:   # the parent re-raises
            out = ('err', f'{type(e).__name__}: {e}', None)
        data = pickle.dumps(out, protocol=4)
        wr.write(struct.pack('<Q', len(data)) + data)
        wr.flush()


if __name__ == '__main__':
    main()
