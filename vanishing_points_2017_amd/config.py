"""Where datasets and model files are looked up -- the attribute names the reference's scripts read from
their config module (config.py:1-8), with the same defaults; VP_DATA_ROOT / VP_CNN_DIR override the two
roots.  caffe_path exists for signature parity only: nothing here imports Caffe."""
import os

_data_root = os.environ.get("VP_DATA_ROOT", "/data/scene_understanding")
_cnn_dir = os.environ.get("VP_CNN_DIR", "./cnn")

caffe_path = ""
yud_path, ecd_path, hlw_path = ("%s/%s" % (_data_root, d) for d in ("YUD", "ECD", "HLW"))
cnn_config_path = "%s/deploy.prototxt" % _cnn_dir
cnn_mean_path = "%s/mean.binaryproto" % _cnn_dir
cnn_weights_path = "%s/weights.caffemodel" % _cnn_dir
