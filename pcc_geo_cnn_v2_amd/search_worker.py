"""Worker process of model_opt.HostSearchPool: reads pickled jobs (length-prefixed) from stdin, runs the host KD-tree part of
the threshold search (model_opt.compute_optimal_thresholds / host_threshold_stats) and writes the pickled result to the pipe
that was its stdout.  Plain numpy / scipy: no GPU, no torch."""
import os
import pickle
import struct
import sys


def main():
    # the binary protocol owns the original stdout; anything printed from here on goes to stderr and cannot corrupt it
    wr = os.fdopen(os.dup(sys.stdout.fileno()), 'wb')
    os.dup2(sys.stderr.fileno(), sys.stdout.fileno())
    sys.stdout = sys.stderr
    from pcc_geo_cnn_v2_amd.model_opt import compute_optimal_thresholds, host_threshold_stats, host_threshold_stats_pruned
    rd = sys.stdin.buffer

    def normals_of(block, with_normals):
        return block[:, block.shape[1] - 3:] if with_normals else None

    while True:
        hdr = rd.read(8)
        if len(hdr) < 8:
            return
        job = pickle.loads(rd.read(struct.unpack('<Q', hdr)[0]))
        try:
            kind, args = (job[0], job[1:]) if isinstance(job[0], str) else ('decide', job)
            if kind == 'tally':
                block, x_hat, thresholds, with_normals = args
                out = ('ok',) + tuple(host_threshold_stats(block, x_hat, thresholds, normals_of(block, with_normals)))
            elif kind == 'tally_pruned':
                block, x_hat, thresholds, with_normals, d1_gpu, resolution, opt_metrics, max_deltas = args
                tallies, mean_tally, kept = host_threshold_stats_pruned(block, x_hat, thresholds, normals_of(block, with_normals), d1_gpu, resolution,
                                                                        opt_metrics, max_deltas)
                out = ('ok', tallies, (mean_tally, kept))
            else:
                block, x_hat, thresholds, resolution, with_normals, opt_metrics, max_deltas = args
                names, best = compute_optimal_thresholds(block, x_hat, thresholds, resolution, normals=normals_of(block, with_normals),
                                                         opt_metrics=opt_metrics, max_deltas=max_deltas, fixed_threshold=False)
                out = ('ok', names, [int(b) for b in best])
        except BaseException as e:   # the parent re-raises
            out = ('err', f'{type(e).__name__}: {e}', None)
        data = pickle.dumps(out, protocol=4)
        wr.write(struct.pack('<Q', len(data)) + data)
        wr.flush()


if __name__ == '__main__':
    main()
