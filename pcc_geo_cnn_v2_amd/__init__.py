"""pcc_geo_cnn_v2_amd -- MI355X (gfx950) native implementation of the 64^3-voxel-block encode+decode
hot path of mauriceqch/pcc_geo_cnn_v2, behind the reference's own Python interface
(ModelConfigType[name].build(), compress/decompress, compress_blocks/decompress_blocks, the
compress_octree.py / decompress_octree.py CLIs and the .ply.bin container)."""
import os as _os

# Four HIP streams carry the codec (kernels and three kinds of copies); the runtime's default of 4 hardware queues makes them
# share a queue as soon as another library (RCCL) opens streams of its own.  Takes effect when set before the HIP runtime loads
# (i.e. before `import torch`); harmless otherwise.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
