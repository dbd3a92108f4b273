#!/bin/bash
# anchor test on the GPU + the structured leg with the sampler's dictionary
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_reference_benchmark.py tests/test_gpu_projection.py -m gpu -q 2>&1 | tail -15
timeout 900 python - <<'P' 2>&1 | tail -40
import json, sys
sys.path.insert(0, '.')
import bench_structured as bs
from kikuchipy_amd import _lib
import inspect
print(inspect.signature(bs.leg))
out = bs.leg(_lib, 0, 10, 64)
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/structured_sampler.json', 'w'), indent=1)
P
