#!/bin/bash
# same-box A/B of an env switch without the test run: bash tools/rounds/r02_ab2.sh VAR [reps]
var=$1
for i in 1 2 ${2:+3}; do for v in 1 0; do env $var=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
