#!/bin/bash
# Developer aid: <pkg>/libqtcnn_ablate.so = the library with conv_pt.hip compiled -DQT_PT_ABLATE (parts of the K loop can be
# switched off at run time: QTCNN_PT_STAGGER_FWD/_BWD = -(bits): 1 weight DMA, 2 patch DMA, 4 weight-fragment reads,
# 8 patch-fragment reads, 16 vmcnt waits, 32 MFMAs).  Results are garbage, only the timing means something:
#   QTCNN_LIB_PATH=$PWD/<pkg>/libqtcnn_ablate.so QTCNN_PT_STAGGER_FWD=-32 python scripts/pt_phases.py
set -e
cd "$(dirname "$0")/.."
PKG=multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd
make -C $PKG/csrc > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DQT_PT_ABLATE \
    -c $PKG/csrc/conv_pt.hip -o /tmp/conv_pt_ablate.o
OBJS=$(ls $PKG/csrc/*.o | grep -v conv_pt.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/conv_pt_ablate.o -o $PKG/libqtcnn_ablate.so
ls -la $PKG/*.so
