#!/bin/bash
# Developer aid: <pkg>/libqtcnn_prof.so = the library with EVERY source compiled -DQT_KERNEL_PROF: conv_igemm / conv_pt then
# carry per-workgroup s_memrealtime phase stamps and export qt_set_igemm_prof / qt_set_pt_prof (scripts/igemm_phases.py,
# scripts/pt_phases.py).  The production library has neither the stamps nor the setters.
#   bash scripts/prof_build.sh && QTCNN_LIB_PATH=$PWD/<pkg>/libqtcnn_prof.so python scripts/pt_phases.py
set -e
cd "$(dirname "$0")/.."
PKG=multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd
OUT=/tmp/qtcnn_prof_objs
mkdir -p $OUT
for f in $PKG/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DQT_KERNEL_PROF \
      -c $f -o $OUT/$(basename ${f%.hip}).o &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/*.o -o $PKG/libqtcnn_prof.so
ls -la $PKG/libqtcnn_prof.so
