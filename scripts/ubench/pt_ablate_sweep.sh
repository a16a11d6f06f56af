L=$PWD/multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd/libqtcnn_ablate.so
for a in 0 1 2 3 4 8 12 16 32 35 44 47 63; do echo "== ablate $a"; QTCNN_LIB_PATH=$L QTCNN_PT_STAGGER_FWD=-$a QTCNN_PT_STAGGER_BWD=-$a timeout -k 10 100 python scripts/pt_phases.py 2>&1 | grep -E "fwd plain" | cut -c1-160; done
