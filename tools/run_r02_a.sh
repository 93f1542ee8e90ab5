#!/bin/bash
# round 2, call A: parity of the sorting-network phase 2 + first 10M bench line on the box, A/B vs the bubble insert
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -25 > $OUT/r02a_pytest.log
tail -3 $OUT/r02a_pytest.log
timeout 600 python bench.py > $OUT/r02a_bench.json 2> $OUT/r02a_bench.err; tail -2 $OUT/r02a_bench.err
for P in phase2_net=0 filter_mfma=0; do
  timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-secondary --param $P > $OUT/r02a_bench_$P.json 2>> $OUT/r02a_bench.err
done
timeout 300 python bench.py --steps 10 --n 50000000 --extent 10 --k 32 --no-cpu-baseline --no-secondary > $OUT/r02a_bench_50m_k32.json 2>> $OUT/r02a_bench.err
timeout 300 python bench.py --steps 10 --n 50000000 --extent 10 --k 32 --no-cpu-baseline --no-secondary --param phase2_net=0 > $OUT/r02a_bench_50m_k32_net0.json 2>> $OUT/r02a_bench.err
# occupancy variant of the k<=16 network kernel
GSX_EXTRA_FLAGS="-DGSX_NET_WAVES17=5" python 3dgsconverter_amd/build.py > /dev/null 2>> $OUT/r02a_bench.err
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-secondary > $OUT/r02a_bench_w5.json 2>> $OUT/r02a_bench.err
GSX_EXTRA_FLAGS="-DGSX_NET_WAVES17=3" python 3dgsconverter_amd/build.py > /dev/null 2>> $OUT/r02a_bench.err
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-secondary > $OUT/r02a_bench_w3.json 2>> $OUT/r02a_bench.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02a_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("secondary",{}).get("ms_per_step"), d.get("cpu_baseline"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
