"""pcc_geo_cnn_v2_amd -- MI355X (gfx950) native implementation of the 64^3-voxel-block encode+decode
hot path of mauriceqch/pcc_geo_cnn_v2, behind the reference's own Python interface
(ModelConfigType[name].build(), compress/decompress, compress_blocks/decompress_blocks, the
compress_octree.py / decompress_octree.py CLIs and the .ply.bin container)."""


def want_hw_queues(n=8):
    """Four HIP streams carry the codec (kernels and three kinds of copies); the runtime's default of 4 hardware queues makes them
    share a queue as soon as another library (RCCL) opens streams of its own (DESIGN_HISTORY.md section 6).  GPU_MAX_HW_QUEUES only takes
    effect when it is set before the HIP runtime loads, i.e. before `import torch`: the entry points (bench.py, the CLIs) call this
    first thing; importing the package no longer changes the host process' environment.  Returns False (and logs) when HIP is
    already loaded, i.e. when the call came too late to matter."""
    import os
    import sys
    if 'GPU_MAX_HW_QUEUES' in os.environ:
        return True
    if 'torch' in sys.modules:
        import logging
        logging.getLogger(__name__).info('GPU_MAX_HW_QUEUES not set and torch (HIP) already imported: copy streams may share a hardware queue with RCCL')
        return False
    os.environ['GPU_MAX_HW_QUEUES'] = str(n)
    return True


__version__ = '0.1.0'
