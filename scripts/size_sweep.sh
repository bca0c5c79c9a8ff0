for s in "512 512" "8192 512" "8192 8192" "1024 1024" "256 256"; do timeout 120 python scripts/size_probe.py $s 10; done
PROBE_OPTS=direct=0 timeout 120 python scripts/size_probe.py 512 512 10
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "direct_framing or random_conf or extreme or golden" 2>&1 | tail -2
