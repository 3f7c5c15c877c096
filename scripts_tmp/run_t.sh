timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "run8" 2>&1 | tail -15
