python tools/ab_so.py pmlp1 | tail -3
python tools/ab_so.py pmlp2 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q -k "mlp or gemm or linear or vit or cswin or mixer or pa_ or xc" 2>&1 | tail -3
python bench.py --no-cpu --no-strict --steps 10 --warmup 3 --only "Vision" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['dominant_kernel']['shapes'] if d['roofline'].get('dominant_kernel') else '')"
