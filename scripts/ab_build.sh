#!/bin/bash
# Developer aid: build the COMMITTED tree as <pkg>/libqtcnn_prev.so and the working tree as
# <pkg>/libqtcnn_hip.so, so that one gpurun call can time both on the same box:
#   QTCNN_LIB_PATH=/root/repo/<pkg>/libqtcnn_prev.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
PKG=multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd
git stash -q
make -C $PKG/csrc > /dev/null
cp $PKG/libqtcnn_hip.so $PKG/libqtcnn_prev.so
git stash pop -q
make -C $PKG/csrc > /dev/null
ls -la $PKG/*.so
