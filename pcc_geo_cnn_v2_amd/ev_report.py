"""D1 / D2 experiment report without the external MPEG `pc_error` binary -- SURVEY.md §8f row 4.

Replaces the report step of /root/reference/src/ev_experiment.py:139-162 (which shells out to `pc_error`, parses its log
with utils/mpeg_parsing.py:38-75 and writes `report_{d1,d2}.json`): the same keys -- pos_total_size_in_bytes,
pos_bits_per_input_point, input_point_count, d1_mse, d1_psnr (+ d2_mse, d2_psnr with normals) -- are computed by
utils/pc_metric.compute_metrics, the module the encoder itself uses for its `.enc.metric.json`
(compress_octree.py:117-118), so the reference's encoder/decoder consistency check (`|d1_psnr diff| < 0.01`,
ev_experiment.py:158-162) carries over.

    python -m pcc_geo_cnn_v2_amd.ev_report --input_pc a.ply --decoded_pc a.ply.bin.ply --enc_pc a.ply.bin \\
        --resolution 1024 [--input_norm a_n.ply] --output report_d1.json
"""
import argparse
import json
import logging
import os

import numpy as np

from .utils import pc_io
from .utils.pc_metric import compute_metrics

logger = logging.getLogger(__name__)


def build_report(input_pc, decoded_pc, enc_pc, resolution, input_norm=None):
    p1 = pc_io.load_pc(input_pc)
    p2 = pc_io.load_pc(decoded_pc)
    n1 = pc_io.load_normals(input_norm) if input_norm else None
    if n1 is not None:
        assert len(n1) == len(p1), 'normals file must have one normal per input point'
    m = compute_metrics(np.asarray(p1, np.float64)[:, :3], np.asarray(p2, np.float64)[:, :3], resolution - 1, p1_n=n1)
    size = os.stat(enc_pc).st_size
    data = {'pos_total_size_in_bytes': size, 'pos_bits_per_input_point': size * 8 / len(p1), 'input_point_count': len(p1)}
    data.update({k: float(v) for k, v in m.items() if k in ('d1_mse', 'd1_psnr', 'd2_mse', 'd2_psnr')})
    return data


def main():
    logging.basicConfig(level=logging.INFO, format='%(asctime)s.%(msecs)03d %(levelname)s %(module)s - %(funcName)s: %(message)s',
                        datefmt='%Y-%m-%d %H:%M:%S')
    p = argparse.ArgumentParser(prog='ev_report.py', description='D1/D2 report for one decoded point cloud.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--input_pc', required=True, help='Path to input point cloud')
    p.add_argument('--decoded_pc', required=True, help='Path to the decoded point cloud')
    p.add_argument('--enc_pc', required=True, help='Path to the compressed file (its size gives the rate)')
    p.add_argument('--input_norm', default=None, help='Path to input point cloud normals (enables D2)')
    p.add_argument('--resolution', type=int, required=True, help='Voxel grid resolution of the input (peak = resolution - 1)')
    p.add_argument('--output', required=True, help='Report JSON path')
    args = p.parse_args()
    data = build_report(args.input_pc, args.decoded_pc, args.enc_pc, args.resolution, args.input_norm)
    with open(args.output, 'w') as f:
        json.dump(data, f, sort_keys=True, indent=4)
    enc_metric = args.enc_pc + '.enc.metric.json'
    if os.path.exists(enc_metric):                       # ev_experiment.py:158-162
        with open(enc_metric) as f:
            enc = json.load(f)
        if 'd1_psnr' in enc:
            diff = abs(enc['d1_psnr'] - data['d1_psnr'])
            logger.info(f'D1 PSNR diff between encoder and decoder: {diff}')
            assert diff < 0.01, f'encoded {args.enc_pc} with D1 {enc["d1_psnr"]} but decoded {args.decoded_pc} with D1 {data["d1_psnr"]}dB'
    logger.info('Done')


if __name__ == '__main__':
    main()
