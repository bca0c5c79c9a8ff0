timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "istft or golden or round_trip or full_size or random or griffinlim or extreme" 2>&1 | tail -2
for i in 1 2 3; do timeout 120 python scripts/size_probe.py 2048 512 30 istft; done
timeout 120 python scripts/size_probe.py 2048 1024 30 istft; timeout 120 python scripts/size_probe.py 2048 256 30 istft
