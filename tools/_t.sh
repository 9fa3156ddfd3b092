export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "double or da64" 2>&1 | tail -4
python bench.py --no-cpu --workload da --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([(b['block'], b['ms']) for b in d['config']['blocks']])"
