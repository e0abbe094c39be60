mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_stream.py tests/test_gpu_multi_native.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | head -3
timeout 2000 python tools/e2e_c5.py > gpurun_out/r04/e2e_c5_full_b.json 2> gpurun_out/r04/e2e_c5_full_b.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/e2e_c5_full_b.json"))
print("identical", d["outputs_identical"], d["md5_first_64MiB"])
for k, v in d["runs"].items():
    print(k, v["seconds"], [t for t in v["timing"] if "read" in t or "pair kernels" in t or "total" in t])
print(d["parity"])
PY
