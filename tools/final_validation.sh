bash tools/evidence.sh suite
bash tools/evidence.sh kernel-stats
bash tools/evidence.sh pmc-gemm | tail -40
bash tools/evidence.sh traffic --no-second-dtype
bash tools/evidence.sh lines
