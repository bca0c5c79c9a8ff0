for s in "8192 512" "8192 2048" "16384 4096"; do timeout 120 python scripts/size_probe.py $s 10; PROBE_OPTS=direct=0 timeout 120 python scripts/size_probe.py $s 10; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "register_ring or direct_framing or extreme or multi_resolution" 2>&1 | tail -2
