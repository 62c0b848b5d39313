"""python -m pcc_geo_cnn_v2_amd.import_tf_checkpoint --checkpoint_dir <tf1 dir> --model_config c3p --output_dir <dir>
Converts the reference's TF1 checkpoint (TensorBundle) into this package's model.npz -- see tf_checkpoint.py."""
from .tf_checkpoint import main

if __name__ == '__main__':
    main()
