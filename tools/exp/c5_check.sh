set -x
export COPRA_NO_BUILD=1
mkdir -p gpurun_out/r05_c5
timeout 900 python -m pytest tests -q -m gpu -x -k "config5 or riccati or interior" 2>&1 | tail -5 > gpurun_out/r05_c5/tests.txt
timeout 600 python tools/try_config5.py 16384 0 > gpurun_out/r05_c5/try.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05_c5/bench.json 2> gpurun_out/r05_c5/bench.err
cat gpurun_out/r05_c5/tests.txt gpurun_out/r05_c5/try.txt
python -c "import json;d=json.loads(open(\"gpurun_out/r05_c5/bench.json\").read().strip().splitlines()[-1]);print(d[\"value\"],d[\"ms_per_step\"]);print(json.dumps({k:v for k,v in d.get(\"extra\",d).items() if \"config5\" in k},indent=0)[:2500])"
