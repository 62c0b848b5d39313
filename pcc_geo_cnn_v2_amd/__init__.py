"""pcc_geo_cnn_v2_amd -- MI355X (gfx950) native implementation of the 64^3-voxel-block encode+decode
hot path of mauriceqch/pcc_geo_cnn_v2, behind the reference's own Python interface
(ModelConfigType[name].build(), compress/decompress, compress_blocks/decompress_blocks, the
compress_octree.py / decompress_octree.py CLIs and the .ply.bin container)."""
__version__ = '0.1.0'
