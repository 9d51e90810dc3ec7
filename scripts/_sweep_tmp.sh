for c in 40 32 36 28; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --dbg 1=$c > gpurun_out/sw_$c.json 2>/dev/null && python -c "
import json;d=json.load(open('gpurun_out/sw_$c.json'));print($c, d['value'],d['ms_per_step'])" || exit 1; done
