"""Path constants with the reference's names (config.py:1-8).  caffe_path is kept for signature
parity only -- nothing here imports Caffe."""
caffe_path = ""

ecd_path = "/data/scene_understanding/ECD"
yud_path = "/data/scene_understanding/YUD"
hlw_path = "/data/scene_understanding/HLW"

cnn_weights_path = "./cnn/weights.caffemodel"
cnn_mean_path = "./cnn/mean.binaryproto"
cnn_config_path = "./cnn/deploy.prototxt"
