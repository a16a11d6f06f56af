for s in 0 2 4 6 8 12; do echo "== stagger bwd $s"; QTCNN_PT_STAGGER_BWD=$s timeout -k 10 100 python scripts/bench_epilogues.py 2>&1 | grep -E "^H|dgrad \+ res|dgrad \+ relu"; done
for s in 0 2 4 8; do echo "== stagger fwd $s"; QTCNN_PT_STAGGER_FWD=$s timeout -k 10 100 python scripts/bench_epilogues.py 2>&1 | grep -E "^H|fwd"; done
